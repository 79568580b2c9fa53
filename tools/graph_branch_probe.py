"""How does the HIP graph executor run two captured branches?  Replay time of small captured graphs of spin kernels (torch.cuda._sleep):
a long side chain beside a main chain, captured before / after the main chain's nodes, with and without the main chain's own short
fork / join pairs (the shape of the training step: text-tower backward beside the decoder backward with its per-block weight-gradient
launches).  Concurrent branches replay in max(a, b), serialised ones in a + b.
Usage (GPU box): python tools/graph_branch_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

SPIN = 200000  # cycles per node


def node():
    torch.cuda._sleep(SPIN)


def build(order, n_main=24, n_side=24, forks=0, side_breaks=0, pad=0):
    """order: 'side_first' | 'main_first'; forks: every `forks` main nodes a one-node branch on stream B is forked and joined one node later;
    side_breaks: every `side_breaks` side nodes the side chain forks / joins a one-node branch of its own (cuts it into segments)"""
    main, A, B, C = (torch.cuda.Stream() for _ in range(4))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main):
            node()
            A.wait_stream(main)
            P = torch.cuda.Stream() if pad else None
            if pad:
                P.wait_stream(main)

            def side_chain():
                if pad:  # a placeholder branch takes the fork node's second out-edge, the side chain its third
                    with torch.cuda.stream(P):
                        node()
                with torch.cuda.stream(A):
                    for i in range(n_side):
                        node()
                        if side_breaks and i % side_breaks == side_breaks - 1 and i + 1 < n_side:
                            C.wait_stream(A)
                            with torch.cuda.stream(C):
                                node()
                            node()
                            A.wait_stream(C)

            def main_chain():
                pend = False
                for i in range(n_main):
                    if pend:
                        main.wait_stream(B)
                        pend = False
                    node()
                    if forks and i % forks == forks - 1 and i + 1 < n_main:
                        B.wait_stream(main)
                        node()  # main's next kernel first (the order Overlap.defer produces)
                        with torch.cuda.stream(B):
                            node()
                        pend = True
                if pend:
                    main.wait_stream(B)

            if order == "side_first":
                side_chain()
                main_chain()
            elif order == "main_one_first":  # main's next node, then the side chain, then the rest of main (what Overlap.defer produces)
                node()
                side_chain()
                main_chain()
            else:
                main_chain()
                side_chain()
            main.wait_stream(A)
            if pad:
                main.wait_stream(P)
            node()
    return g


def time_graph(g, reps=5):
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    torch.cuda._sleep(SPIN)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        node()
    e1.record()
    torch.cuda.synchronize()
    unit = e0.elapsed_time(e1) / 20
    print(f"one node = {unit * 1e3:.1f} us (eager, back to back)")
    for name, kw in [("two plain chains, side captured first", dict(order="side_first")),
                     ("two plain chains, main captured first", dict(order="main_first")),
                     ("main forks every 4 nodes, side first", dict(order="side_first", forks=4)),
                     ("main forks every 4 nodes, main first", dict(order="main_first", forks=4)),
                     ("main forks every 4, one main node then side", dict(order="main_one_first", forks=4)),
                     ("the same with a placeholder second edge", dict(order="main_one_first", forks=4, pad=1)),
                     ("main first + placeholder", dict(order="main_first", forks=4, pad=1)),
                     ("main only (no side chain)", dict(order="main_first", n_side=0, forks=4))]:
        try:
            t = time_graph(build(**kw))
            print(f"{name:55s}: {t * 1e3:8.1f} us = {t / unit:5.1f} nodes")
        except Exception as e:
            print(f"{name:55s}: failed: {str(e)[:120]}")


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def contention_probe():
    """two independent captured chains under CONTENTION: the main chain = chip-filling GEMMs back to back, the side chain = small GEMMs;
    does the side chain make progress while the main chain always has a ready kernel?"""
    dev = "cuda"
    a = torch.randn(8192, 2048, device=dev, dtype=torch.bfloat16)
    b = torch.randn(2048, 8192, device=dev, dtype=torch.bfloat16)
    c = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
    sa = torch.randn(512, 512, device=dev, dtype=torch.bfloat16)
    sc = torch.empty(512, 512, device=dev, dtype=torch.bfloat16)
    torch.mm(a, b, out=c)
    torch.mm(sa, sa, out=sc)
    torch.cuda.synchronize()

    def cap(n_main, n_side, side_prio=0):
        main, A = torch.cuda.Stream(), torch.cuda.Stream(priority=side_prio)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(main):
            with torch.cuda.graph(g, stream=main):
                torch.mm(sa, sa, out=sc)
                A.wait_stream(main)
                with torch.cuda.stream(A):
                    for _ in range(n_side):
                        torch.mm(sa, sa, out=sc)
                for _ in range(n_main):
                    torch.mm(a, b, out=c)
                main.wait_stream(A)
        return g

    for ns in (50, 100, 200):
        print(f"  40 big + {ns} small: {time_graph(cap(40, ns)) * 1e3:.0f} us (small alone {time_graph(cap(0, ns)) * 1e3:.0f})")
    tm, ts = time_graph(cap(40, 0)), time_graph(cap(0, 400))
    tb = time_graph(cap(40, 400))
    tp = time_graph(cap(40, 400, side_prio=-1))
    print(f"contention: 40 big GEMMs alone {tm * 1e3:.0f} us, 400 small GEMMs alone {ts * 1e3:.0f} us, both {tb * 1e3:.0f} us "
          f"(max = concurrent, sum = serialised), side chain captured on a high-priority stream {tp * 1e3:.0f} us")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "contention":
    contention_probe()
