"""Summarise a DEBUG_HIP_GRAPH_DOT_PRINT=1 dump of the captured step: per graph stream id the number of nodes, and every edge that
crosses stream ids (the cross-queue waits of the replay).  Usage: python tools/graph_dot_summary.py <dot file> [kernel substring ...]"""
import re
import sys
from collections import Counter, defaultdict


def load(path):
    txt = open(path).read()
    nodes = {}
    for m in re.finditer(r'"graph_1_node_(\d+)"\[[^\]]*label="(\d+)\n(.*?)\nStreamId:(\d+)\nSignalIsRequired: (\w+)', txt, re.S):
        nodes[int(m.group(1))] = dict(name=m.group(3).replace("\n", " "), stream=int(m.group(4)), signal=m.group(5) == "true")
    edges = [(int(a), int(b)) for a, b in re.findall(r'"graph_1_node_(\d+)"\s*->\s*"graph_1_node_(\d+)"', txt)]
    return nodes, edges


def short(n):
    n = re.sub(r"^_ZN3vtp\d+", "", n)
    n = re.sub(r"^_ZN2at6native\d+", "at::", n)
    return n[:44]


def main():
    nodes, edges = load(sys.argv[1])
    print(len(nodes), "nodes,", len(edges), "edges; nodes per stream id:", dict(sorted(Counter(v["stream"] for v in nodes.values()).items())))
    preds, succs = defaultdict(list), defaultdict(list)
    for a, b in edges:
        preds[b].append(a)
        succs[a].append(b)
    cross = [(a, b) for a, b in edges if nodes[a]["stream"] != nodes[b]["stream"]]
    print(len(cross), "cross-stream edges")
    for pat in sys.argv[2:]:
        print("== nodes matching", pat)
        for i, v in sorted(nodes.items()):
            if pat in v["name"]:
                print(f"  #{i} s{v['stream']} sig={int(v['signal'])} {short(v['name'])}  <- " +
                      ", ".join(f"#{p}(s{nodes[p]['stream']} {short(nodes[p]['name'])[:18]})" for p in preds[i]) + "  -> " +
                      ", ".join(f"#{c}(s{nodes[c]['stream']})" for c in succs[i]))
    # runs of consecutive node ids per stream (capture order)
    runs, cur = [], None
    for i in sorted(nodes):
        s = nodes[i]["stream"]
        if cur and cur[0] == s:
            cur[2] = i
        else:
            cur = [s, i, i]
            runs.append(cur)
    print("capture-order runs (stream: first..last id) with >= 12 nodes:")
    for s, a, b in runs:
        if b - a >= 11:
            print(f"  s{s}: #{a}..#{b}  {short(nodes[a]['name'])} .. {short(nodes[b]['name'])}")


if __name__ == "__main__":
    main()
