"""What a kernel that HOLDS CUs for a long time (RCCL channels during the backward at N > 1; another stream's one-workgroup-per-CU kernel)
costs the persistent GEMMs -- the one-GPU proxy for the 8-GPU step.  A do-nothing kernel (vtp_cu_thief) pins 0 / 16 /
32 CUs on a side stream while a chain of the step's NT GEMMs runs on the main stream; VTP_GEMM_DYN=0 / 1 (static tile lists / tiles
drawn from per-XCD queues) and VTP_GEMM_CUS=n (persistent grid capped to n workgroups) are compared in separate processes.
Usage (GPU box): VTP_GEMM_DYN=1 python tools/cu_thief.py [tag]"""
import os

os.environ.setdefault("VTP_DIAG", "1")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtp_amd import _lib, ops

CHAIN = [  # the trunk block's forward chain at 34 144 rows: (M, N, K, kind)
    (34144, 2304, 768, "bf16"), (34144, 768, 768, "f32"), (34144, 4096, 768, "swiglu"), (34144, 768, 2048, "f32"),
    (34144, 2048, 768, "bf16"), (34144, 768, 4096, "bf16"), (34144, 768, 2304, "bf16"), (34144, 768, 768, "bf16"),
]


def serve(ncu: int, seconds: float):
    """hold `ncu` CUs from THIS process for `seconds` (back-to-back 20-ms launches): the thief of a step-level experiment has to live
    in another process -- inside the bench process its stream shares a hardware queue with streams of the step, and a kernel that
    sits in a queue for 40 ms serialises everything behind it there (measured: the step took exactly thief + step)"""
    import time
    lib = _lib.load()
    st = torch.cuda.Stream()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        _lib.check(lib.vtp_cu_thief(ncu, 2000000, None, st.cuda_stream), "vtp_cu_thief")
        n += 1
        if n % 4 == 0:
            st.synchronize()  # keep at most a few launches queued
    st.synchronize()
    print(f"[thief] held {ncu} CUs for {time.time() - t0:.1f} s ({n} launches)", flush=True)


def main():
    if len(sys.argv) > 3 and sys.argv[1] == "--serve":
        return serve(int(sys.argv[2]), float(sys.argv[3]))
    tag = sys.argv[1] if len(sys.argv) > 1 else f"dyn={os.environ.get('VTP_GEMM_DYN', '1')} cus={os.environ.get('VTP_GEMM_CUS', '-')}"
    lib = _lib.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    fns = []
    for M, N, K, kind in CHAIN:
        a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g)
        if kind == "f32":
            c = torch.zeros(M, N, device=dev)
            fns.append(lambda a=a, w=w, c=c, bias=bias, M=M, N=N, K=K: ops.gemm_nt(a, w, c, M=M, N=N, K=K, bias=bias, resid=c, epi=ops.EPI_F32))
        elif kind == "swiglu":
            c = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
            c2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            fns.append(lambda a=a, w=w, c=c, c2=c2, bias=bias, M=M, N=N, K=K: ops.gemm_nt(a, w, c, M=M, N=N, K=K, bias=bias, c2=c2, epi=ops.EPI_SWIGLU))
        else:
            c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            fns.append(lambda a=a, w=w, c=c, bias=bias, M=M, N=N, K=K: ops.gemm_nt(a, w, c, M=M, N=N, K=K, bias=bias, epi=ops.EPI_BF16))

    def chain():
        for f in fns:
            f()

    side = torch.cuda.Stream()
    for _ in range(3):
        chain()
    torch.cuda.synchronize()
    base = None
    for ncu in (0, 16, 32, 64):
        ts = []
        for rep in range(7):
            torch.cuda.synchronize()
            if ncu:
                with torch.cuda.stream(side):
                    _lib.check(lib.vtp_cu_thief(ncu, 400000, None, side.cuda_stream), "vtp_cu_thief")  # 4 ms: longer than the chain
            torch.cuda._sleep(400000)  # clocks up (and the thief resident) before the chain starts -- the same preamble without a thief
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            chain()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        med = ts[len(ts) // 2]
        base = med if ncu == 0 else base
        print(f"[{tag}] {ncu:3d} CUs held: chain of {len(fns)} GEMMs {med:8.1f} us  x{med / base:.3f} of the free chip (CU share alone: x{256 / (256 - ncu):.3f})", flush=True)


if __name__ == "__main__":
    main()
