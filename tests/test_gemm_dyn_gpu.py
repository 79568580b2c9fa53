"""Dynamic tile assignment of the persistent NT GEMMs (vtp_amd/csrc/gemm8p.hip gemm8p_body<.., DYN>, and the same queues in gemm8h.hip /
gemm4w.hip: workgroups DRAW their tiles from per-XCD queues in device memory instead of owning the static list bx, bx + G, ...;
vtp_set_gemm_dynamic).  Which workgroup computes
a tile cannot change a single bit of it -- every comparison is BIT FOR BIT against the static launch: every epilogue of the step, ragged
M tails, launches back to back on one stream (the queue words must be back at zero), on two streams at once (one queue slot per
stream), and with CUs taken away while the launch runs (a do-nothing kernel holds 32 of them: the case the mechanism exists for)."""
import os

import pytest
import torch

from test_kernels_gpu import DEV, bf, ops  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if os.environ.get("VTP_GEMM_DYN") is not None:
        pytest.skip("VTP_GEMM_DYN in the environment overrides vtp_set_gemm_dynamic")
    yield
    from vtp_amd import _lib
    _lib.load().vtp_set_gemm_dynamic(0)
    _lib.load().vtp_set_gemm_tuning(-1, 3)


def _dyn(on):
    from vtp_amd import _lib
    _lib.check(_lib.load().vtp_set_gemm_dynamic(int(on)), "vtp_set_gemm_dynamic")


def _launchers(M, N, K, kind, g):
    o = ops()
    a = bf(torch.randn(M, K, device=DEV, generator=g))
    w = bf(torch.randn(N, K, device=DEV, generator=g) * 0.05)
    bias = torch.randn(N, device=DEV, generator=g)
    if kind == "bf16":
        outs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)]
        run = lambda: o.gemm_nt(a, w, outs[0], M=M, N=N, K=K, bias=bias, epi=o.EPI_BF16)
    elif kind == "f32res":
        base = torch.randn(M, N, device=DEV, generator=g)
        outs = [torch.empty(M, N, device=DEV)]
        run = lambda: o.gemm_nt(a, w, outs[0], M=M, N=N, K=K, bias=bias, resid=base, epi=o.EPI_F32)
    elif kind == "swiglu":
        outs = [torch.full((M, N // 2), float("nan"), dtype=torch.bfloat16, device=DEV), torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)]
        run = lambda: o.gemm_nt(a, w, outs[0], M=M, N=N, K=K, bias=bias, c2=outs[1], epi=o.EPI_SWIGLU)
    elif kind == "gelu":
        outs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV), torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)]
        run = lambda: o.gemm_nt(a, w, outs[0], M=M, N=N, K=K, bias=bias, c2=outs[1], epi=o.EPI_GELU)
    else:  # fused apply_rope in the qkv projection
        outs = [torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)]
        pos = (torch.arange(M, dtype=torch.int32, device=DEV) % 257 - 1)
        sin, cos = (bf(torch.randn(256, 64, device=DEV, generator=g)) for _ in range(2))
        run = lambda: o.gemm_qkv_rope(a, w, bias, outs[0], M, N, K, pos, sin, cos, 2 * (N // 3))
    return run, outs


SHAPES = [(8, 34144, 768, 768, "f32res"), (8, 34144, 2304, 768, "rope"), (8, 34144, 4096, 768, "swiglu"), (8, 34144, 768, 2048, "f32res"),
          (8, 34144, 768, 768, "bf16"), (8, 16448, 2304, 768, "bf16"), (8, 70001, 768, 256, "bf16"), (8, 33000, 1536, 320, "gelu"),
          # the half-size kernel (two workgroups per CU: 512 slots) and the one-wave-per-SIMD kernel, forced
          (9, 34144, 2304, 768, "rope"), (9, 34144, 2048, 768, "bf16"), (9, 34144, 4096, 768, "swiglu"), (9, 70001, 768, 768, "f32res"),
          (9, 40000, 1024, 192, "bf16"), (10, 34144, 768, 4096, "bf16"), (10, 34144, 768, 2304, "bf16"), (10, 70001, 512, 2048, "bf16")]


@pytest.mark.parametrize("cfg,M,N,K,kind", SHAPES)
def test_dynamic_tiles_bit_identical_to_static(cfg, M, N, K, kind):
    from vtp_amd import _lib
    _lib.load().vtp_set_gemm_tuning(cfg, 3)  # one kernel for the shape (the dispatch would send it where it measured best)
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    run, outs = _launchers(M, N, K, kind, g)
    _dyn(False)
    run()
    ref = [t.clone() for t in outs]
    for t in outs:
        t.fill_(float("nan"))
    _dyn(True)
    for rep in range(3):  # back to back: the last workgroup of a launch leaves the queue words at zero for the next one
        run()
        torch.cuda.synchronize()
        for t, r in zip(outs, ref):
            assert torch.equal(t.view(torch.int16 if t.dtype == torch.bfloat16 else torch.int32), r.view(torch.int16 if r.dtype == torch.bfloat16 else torch.int32)), \
                f"{kind} {M}x{N}x{K} launch {rep}: {int((t != r).sum())} elements differ from the static launch"
        if rep < 2:
            for t in outs:
                t.fill_(float("nan"))


def test_dynamic_tiles_two_streams_and_missing_cus():
    """two persistent launches in flight on two streams (each stream has its own queue slot), then the same beside a kernel that holds
    32 CUs for the whole time: results bit-identical to the static launches on an idle chip"""
    from vtp_amd import _lib
    lib = _lib.load()
    lib.vtp_set_gemm_tuning(8, 3)
    g = torch.Generator(device=DEV).manual_seed(5)
    r1, o1 = _launchers(34144, 2304, 768, "bf16", g)
    r2, o2 = _launchers(34144, 768, 768, "f32res", g)
    _dyn(False)
    r1()
    r2()
    torch.cuda.synchronize()
    ref1, ref2 = o1[0].clone(), o2[0].clone()
    _dyn(True)
    s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    thief = os.environ.get("VTP_DIAG") == "1"
    for held in ((0, 32) if thief else (0,)):
        o1[0].fill_(float("nan"))
        o2[0].fill_(float("nan"))
        torch.cuda.synchronize()
        if held:
            with torch.cuda.stream(s3):
                _lib.check(lib.vtp_cu_thief(held, 300000, None, s3.cuda_stream), "vtp_cu_thief")  # 3 ms
        for rep in range(4):
            with torch.cuda.stream(s1):
                r1()
            with torch.cuda.stream(s2):
                r2()
        torch.cuda.synchronize()
        assert torch.equal(o1[0].view(torch.int16), ref1.view(torch.int16)), f"stream 1, {held} CUs held"
        assert torch.equal(o2[0].view(torch.int32), ref2.view(torch.int32)), f"stream 2, {held} CUs held"
