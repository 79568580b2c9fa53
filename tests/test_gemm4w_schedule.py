"""Static checks of the GENERATED k loops of the one-wave-per-SIMD GEMMs (tools/gen_gemm4w_ktile.py -> vtp_amd/csrc/gemm4w_ktile.inc,
gemm4w_tn_ktile.inc).  The loops are inline asm: hipcc inserts no s_waitcnt, sees no LDS hazards and counts no LDS-DMA pieces in them,
so the three things it would normally guarantee are re-checked here by a small abstract interpreter that runs the instruction list of
the loop body (several iterations, as the hardware would):

  1. register data hazards -- an MFMA (or a column-sum v_dot2) never reads a fragment register with a ds_read still outstanding on it:
     every read is retired by an `s_waitcnt ... lgkmcnt(0)` first.  The same rule catches a prefetch that overwrites fragments the
     current step still multiplies (the later MFMAs would read a pending register).
  2. LDS hazards between the waves -- a region of a ring slot (A rows lo / hi, B) is overwritten by LDS-DMA only behind a barrier that
     follows the last fragment read of that region (with the reads retired in front of the barrier: each wave waits for its own), and
     is read again only behind an `s_waitcnt vmcnt(N)` that retires those pieces (the wave's own; N counts the younger ones) plus a
     barrier (everybody else's).
  3. bookkeeping -- 64 MFMAs per k-tile covering every (accumulator, k-step) exactly once in ascending k-step order (the order the
     8-phase kernel uses: bit identity), 16 LDS-DMA pieces per wave and k-tile with M0 written at least two instructions ahead of the
     load that uses it, M0 saved and restored.

CPU only: the GPU tests check the numbers, this checks that they are not right by luck of timing."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen4w", os.path.join(ROOT, "tools", "gen_gemm4w_ktile.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return gen


def _regs(op):
    """register names an operand text stands for: an asm operand %[name] is one opaque register (group); v[a:b] expands"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    if m:
        return {f"v{i}" for i in range(int(m.group(1)), int(m.group(2)) + 1)}
    return {op}


class Machine:
    """region keys: (slot, 'A-lo' | 'A-hi' | 'B'); fragment reads and DMA pieces are mapped onto them by `classify_*`"""

    def __init__(self, classify_read, classify_dma):
        self.pending = set()           # registers with an outstanding ds_read
        self.cread, self.cdma = classify_read, classify_dma
        self.read_open = {}            # region -> "reads issued, not yet retired + barriered" (True) since the last barrier
        self.read_retired = set()      # regions whose reads were retired by lgkmcnt(0) but no barrier yet
        self.dma_age = []              # regions of the outstanding pieces, oldest first
        self.unlanded = {}             # region -> number of pieces issued and not yet (retired by vmcnt AND barriered)
        self.retired_not_barriered = {}
        self.m0_age = None             # instructions since the last M0 write
        self.mfma = []                 # (acc, k-step) in program order, per k-tile
        self.n_dma = 0
        self.errors = []

    def err(self, i, ins, msg):
        self.errors.append(f"#{i} `{ins}`: {msg}")

    def step(self, i, ins):
        t = ins.replace(",", " ").split()
        op = t[0]
        if self.m0_age is not None:
            self.m0_age += 1
        if op.startswith("ds_read"):
            dst, addr = t[1], t[2]
            off = int(t[3].split(":")[1]) if len(t) > 3 else 0
            region = self.cread(addr, off)
            if self.unlanded.get(region, 0) or self.retired_not_barriered.get(region, 0):
                self.err(i, ins, f"reads {region} while LDS-DMA pieces into it are not retired + barriered")
            self.pending |= _regs(dst)
            self.read_open[region] = True
        elif op == "v_mfma_f32_32x32x16_bf16":
            for src in (t[2], t[3]):
                if _regs(src) & self.pending:
                    self.err(i, ins, f"reads {src} with a ds_read outstanding on it")
            self.mfma.append((t[1], t[2], t[3]))
        elif op == "v_dot2_f32_bf16":
            if _regs(t[2]) & self.pending:
                self.err(i, ins, f"reads {t[2]} with a ds_read outstanding on it")
        elif op == "s_waitcnt":
            txt = " ".join(t[1:])
            if "lgkmcnt(0)" in txt:
                self.pending.clear()
                self.read_retired |= {r for r, o in self.read_open.items() if o}
            m = re.search(r"vmcnt\((\d+)\)", txt)
            if m:
                keep = int(m.group(1))
                while len(self.dma_age) > keep:
                    for r in self.dma_age.pop(0):
                        self.unlanded[r] -= 1
                        self.retired_not_barriered[r] = self.retired_not_barriered.get(r, 0) + 1
        elif op == "s_barrier":
            if self.pending:
                self.err(i, ins, "barrier with fragment reads outstanding (the other waves may overwrite what they read)")
            for r in self.read_retired:
                self.read_open[r] = False
            self.read_retired.clear()
            self.retired_not_barriered.clear()
        elif op == "global_load_lds_dwordx4":
            if self.m0_age is None or self.m0_age < 2:
                self.err(i, ins, "M0 written less than two instructions ahead of the LDS-DMA load")
            regions = self.cdma(self.m0_src)  # every region the piece's image overlaps
            for region in regions:
                if self.read_open.get(region):
                    self.err(i, ins, f"overwrites {region} whose fragment reads are not retired + barriered")
                self.unlanded[region] = self.unlanded.get(region, 0) + 1
            self.dma_age.append(regions)
            self.n_dma += 1
        elif op in ("s_mov_b32", "s_add_u32") and t[1] == "m0":
            self.m0_age = 0
            if op == "s_mov_b32":
                self.m0_src = t[2]


def _run(body, classify_read, classify_dma, iterations=3):
    head = body.index("1:")
    tail = next(k for k, l in enumerate(body) if l.startswith("s_cbranch"))
    m = Machine(classify_read, classify_dma)
    assert body[0] == "s_mov_b32 %[sm], m0" and body[-1] == "s_mov_b32 m0, %[sm]", "M0 must be saved and restored"
    k = 0
    for l in body[1:head]:
        m.step(k, l)
        k += 1
    per_iter = []
    for _ in range(iterations):
        m.mfma, m.n_dma = [], 0
        for l in body[head + 1:tail]:
            m.step(k, l)
            k += 1
        per_iter.append((list(m.mfma), m.n_dma))
    for l in body[tail + 1:]:
        m.step(k, l)
        k += 1
    assert not m.pending, "fragment reads outstanding when the asm statement ends"
    return m, per_iter


def _check_mfma_order(mfmas, acc_of):
    """two k-tiles per iteration: every accumulator gets k-steps 0, 1, 2, 3 of each k-tile once, in ascending order"""
    assert len(mfmas) == 128
    for half in (mfmas[:64], mfmas[64:]):
        seen = {}
        for acc, b, a in half:
            ks = acc_of(acc, b, a)
            seen.setdefault(acc, []).append(ks)
        assert len(seen) == 16 and all(v == [0, 1, 2, 3] for v in seen.values()), seen


def test_nt_schedule_is_hazard_free():
    gen = _gen()

    def cread(addr, off):  # %[x{slot}{ks}] = A image (row blocks 0,1 lo / 2,3 hi), %[y{slot}{ks}] = B image
        kind, slot = addr[2], int(addr[3])
        return (slot, "B") if kind == "y" else (slot, "A-lo" if off < 8192 else "A-hi")

    def cdma(src):  # %[da{slot}] / %[db{slot}]: the waves' shares of the A / B images (between them: rows lo AND hi)
        slot = int(src[4])
        return [(slot, "A-lo"), (slot, "A-hi")] if src[3] == "a" else [(slot, "B")]

    body = gen.body()
    m, per_iter = _run(body, cread, cdma)
    assert not m.errors, "\n".join(m.errors[:10])

    def ks_of(acc, b, a):  # fragment operand %[p|q n] : n = ks * 4 + i ; %[l|h n] : n = ks * 2 + j
        return int(re.search(r"\d+", b).group()) // 4

    for mf, ndma in per_iter:
        _check_mfma_order(mf, ks_of)
        assert ndma == 32  # 16 pieces per wave and k-tile


def test_tn_schedule_is_hazard_free():
    gen = _gen()

    def cread(addr, off):  # %[a{E|O}{slot}] = A sub-images (lo: offset < 8192, hi: + 8192), %[b{E|O}{slot}] = B sub-images
        kind, slot = addr[2], int(addr[4])
        return (slot, "B") if kind == "b" else (slot, "A-lo" if off < 8192 else "A-hi")

    def cdma(src):
        slot = int(src[4])
        return [(slot, "A-lo"), (slot, "A-hi")] if src[3] == "a" else [(slot, "B")]

    for csum in (False, True):
        body = gen.tn_body(csum)
        m, per_iter = _run(body, cread, cdma)
        assert not m.errors, "\n".join(m.errors[:10])

        def ks_of(acc, b, a):  # B fragment v[r:r+3]: r = 128 | 192 + 4 * (ks * 4 + i)
            r = int(re.match(r"v\[(\d+):", b).group(1))
            return ((r - 128) % 64) // 16

        for mf, ndma in per_iter:
            _check_mfma_order(mf, ks_of)
            assert ndma == 32
        if csum:  # every A fragment register of a k-tile is summed exactly once per k-tile
            loop = body[body.index("1:") + 1:]
            dots = [l.replace(",", " ").split()[2] for l in loop if l.startswith("v_dot2")]
            assert len(dots) == 128 and len(set(dots)) == 64


def test_fragment_register_map_of_the_tn_loop_is_disjoint():
    """physical fragment registers: l 64..95, h 96..127, p 128..191, q 192..255 -- the ranges the clobber list of gemm4w_tn.hip names"""
    gen = _gen()
    seen = {}
    for kind, n in (("l", 8), ("h", 8), ("p", 16), ("q", 16)):
        for i in range(n):
            r = gen.FR(kind, i)
            for w in range(4):
                assert r + w not in seen, (kind, i, seen[r + w])
                seen[r + w] = (kind, i)
    assert sorted(seen) == list(range(64, 256))
    inc = open(os.path.join(ROOT, "vtp_amd", "csrc", "gemm4w_tn_ktile.inc")).read()
    clob = re.search(r"#define W4T_TILE_CLOBBERS (.*)", inc).group(1)
    assert [int(x) for x in re.findall(r'"v(\d+)"', clob)] == list(range(64, 256))


def test_the_checker_catches_seeded_hazards():
    """the interpreter above is not vacuous: four one-line mutations of the NT loop, each a real bug class, are all flagged"""
    gen = _gen()

    def cread(addr, off):
        kind, slot = addr[2], int(addr[3])
        return (slot, "B") if kind == "y" else (slot, "A-lo" if off < 8192 else "A-hi")

    def cdma(src):
        slot = int(src[4])
        return [(slot, "A-lo"), (slot, "A-hi")] if src[3] == "a" else [(slot, "B")]

    body = gen.body()
    errs = lambda b: _run(b, cread, cdma)[0].errors
    assert not errs(body)
    b = list(body)
    del b[b.index("s_barrier")]                                    # no barrier in front of the X step's staging
    assert any("overwrites" in e for e in errs(b))
    assert any("not retired" in e for e in errs([l.replace("vmcnt(8)", "vmcnt(16)") for l in body]))       # pieces still in flight
    assert any("barrier with fragment reads outstanding" in e
               for e in errs(["s_waitcnt vmcnt(8)" if l == "s_waitcnt vmcnt(8) lgkmcnt(0)" else l for l in body]))
    assert any("overwrites" in e for e in errs([l.replace("%[db0]", "%[da0]") for l in body]))            # A refilled a step early
    b = list(body)
    k = next(i for i, l in enumerate(b) if l.startswith("ds_read_b128 %[h0]"))
    b.insert(k + 1, "v_mfma_f32_32x32x16_bf16 %[c00], %[p0], %[h0], %[c00]")                               # use before the wait
    assert any("ds_read outstanding" in e for e in errs(b))
