#!/usr/bin/env python3
"""Generates vtp_amd/csrc/gemm4w_ktile.inc: the hand-scheduled k loop of ONE output tile of the one-wave-per-SIMD GEMM (gemm4w.hip) as
a single inline-asm statement (hipcc cannot hold 256 accumulator AGPRs + ~200 fragment VGPRs through a loop it allocates itself: it
moves / spills hundreds of registers per k-tile -- see the header of gemm4w.hip).

Wave tile 128 x 128 = acc[i][j], i = B column block, j = A row block (4 x 4 accumulators of v_mfma_f32_32x32x16_bf16).  A k-tile (64
deep = k-steps 0..3) of ring slot c is multiplied in two steps of 32 MFMAs split by OUTPUT ROWS, so that most of the slot is released
one step before the k-tile's last MFMA and the LDS-DMA refill is spread over both steps (the CU's texture-address path moves 64 B/clk:
the 64 KiB of a k-tile keep it busy for half of the k-tile's 2048 MFMA cycles -- issued inside one step the waves stall on it):
  S1   s_waitcnt lgkmcnt(0) ; s_barrier          every wave has fetched A-lo and B of this k-tile: those LDS regions are free
  X    MFMAs rows 0..63 (A-lo x B, 4 k-steps); fetch A-hi of this k-tile (8 ds_read_b128, groups 0..7); stage the wave's 8 pieces of
       the B images of the k-tile after next into this slot (groups 0, 2, .. 14); staging cursor decision + B cursor advance
  S2   s_waitcnt vmcnt(8) lgkmcnt(0) ; s_barrier  A-hi fetched by every wave; everything but the 8 pieces just issued has landed
  Y    MFMAs rows 64..127 (A-hi x B); fetch A-lo and B of the NEXT k-tile from the other slot (24 reads, groups 0..11, B into the other
       B register set); stage the wave's 8 pieces of the A images (groups 0, 2, .. 14); A cursor advance
The loop body is two k-tiles (slot 0 with B set p, slot 1 with B set q): nk must be even.
Fragment indices: A-lo l[ks*2 + j], A-hi h[ks*2 + (j-2)], B p|q[ks*4 + i].  Read addresses x{slot}{ks} (A image), y{slot}{ks} (B image).
Run: python tools/gen_gemm4w_ktile.py  (rewrites the .inc; the file is committed, the build does not run this)."""
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vtp_amd", "csrc", "gemm4w_ktile.inc")
DMA_GROUPS = [0, 2, 4, 6, 8, 10, 12, 14]


def dma_m0(L, which, n, slot):
    """M0 = LDS destination of piece n, written at the HEAD of the group (two MFMAs ahead of the load that reads it, and never right
    behind the previous piece's load)"""
    L.append(f"s_mov_b32 m0, %[d{which}{slot}]" if n == 0 else "s_add_u32 m0, m0, 0x400")
    ptr, t = (f"po{which}", "vt1") if n & 1 else (f"pe{which}", "vt0")
    L.append(f"v_min_u32 %[{t}], %[{ptr}], %[vmax{which}]")  # (the clamped lane offset, also ahead of the load)


def dma_piece(L, which, n, slot):
    """piece n (0..7) of the wave's share of operand `which` ('a' | 'b') into ring slot `slot`"""
    ptr, t = (f"po{which}", "vt1") if n & 1 else (f"pe{which}", "vt0")
    L.append(f"global_load_lds_dwordx4 %[{t}], %[mat{which}]")
    L.append(f"v_add_u32 %[{ptr}], %[step{which}], %[{ptr}]")


def ktile(L, slot, bc, bn, diag):
    o = 1 - slot
    # ---- S1
    L += ["s_waitcnt lgkmcnt(0)"] + ([] if diag == 2 else ["s_barrier"])
    # ---- X: rows 0..63
    reads = [(f"h{ks * 2 + b}", f"x{slot}{ks}", (2 + b) * 4096) for ks in range(4) for b in range(2)]
    for g in range(16):
        ks, q = g >> 2, g & 3
        j, i0 = q >> 1, (q & 1) * 2
        if g in DMA_GROUPS and diag != 3:
            dma_m0(L, "b", DMA_GROUPS.index(g), slot)
        for i in (i0, i0 + 1):
            L.append(f"v_mfma_f32_32x32x16_bf16 %[c{i}{j}], %[{bc}{ks * 4 + i}], %[l{ks * 2 + j}], %[c{i}{j}]")
            if i == i0 and g < 8:  # (fragment reads in the first MFMA's shadow, the LDS-DMA piece in the second's)
                d, a, off = reads[g]
                L.append(f"ds_read_b128 %[{d}], %[{a}] offset:{off}")
        if g in DMA_GROUPS and diag != 3:
            dma_piece(L, "b", DMA_GROUPS.index(g), slot)
        if g == 15:  # staging cursor: does it leave its output tile after this k-tile?  (cc = k-tiles left in the cursor's tile)
            L += ["s_sub_u32 %[cc], %[cc], 1", "s_cmp_eq_u32 %[cc], 0", "s_cselect_b32 %[sadvb], %[tadvb], %[kadvb]",
                  "s_cselect_b32 %[sadva], %[tadva], %[kadva]", "s_cselect_b32 %[cc], %[nkr], %[cc]",
                  "v_add_u32 %[peb], %[sadvb], %[peb]", "v_add_u32 %[pob], %[sadvb], %[pob]"]
    # ---- S2
    L += [["s_waitcnt vmcnt(8) lgkmcnt(0)", "s_barrier"], ["s_waitcnt lgkmcnt(0)", "s_barrier"], ["s_waitcnt lgkmcnt(0)"],
          ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]][diag]
    # ---- Y: rows 64..127; fetch A-lo + B of the next k-tile (other slot)
    reads = []
    for ks in range(4):
        reads += [(f"l{ks * 2 + b}", f"x{o}{ks}", b * 4096) for b in range(2)]
        reads += [(f"{bn}{ks * 4 + i}", f"y{o}{ks}", i * 4096) for i in range(4)]
    for g in range(16):
        ks, q = g >> 2, g & 3
        j, i0 = q >> 1, (q & 1) * 2
        if g in DMA_GROUPS and diag != 3:
            dma_m0(L, "a", DMA_GROUPS.index(g), slot)
        for i in (i0, i0 + 1):
            L.append(f"v_mfma_f32_32x32x16_bf16 %[c{i}{2 + j}], %[{bc}{ks * 4 + i}], %[h{ks * 2 + j}], %[c{i}{2 + j}]")
            if i == i0:
                for d, a, off in reads[2 * g:2 * g + 2] if g < 12 else []:
                    L.append(f"ds_read_b128 %[{d}], %[{a}] offset:{off}")
        if g in DMA_GROUPS and diag != 3:
            dma_piece(L, "a", DMA_GROUPS.index(g), slot)
        if g == 15:
            L += ["v_add_u32 %[pea], %[sadva], %[pea]", "v_add_u32 %[poa], %[sadva], %[poa]"]


def body(diag=0):
    """diag (timing experiments only, WRONG results): 1 = no vmcnt wait at S2 | 2 = no barriers either | 3 = no LDS-DMA"""
    L = ["s_mov_b32 %[sm], m0", "1:"]
    ktile(L, 0, "p", "q", diag)
    ktile(L, 1, "q", "p", diag)
    L += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 1b", "s_waitcnt lgkmcnt(0)", "s_mov_b32 m0, %[sm]"]
    return L


def emit(name, lines):
    out = [f"#define {name} \\"]
    for l in lines:
        out.append(f'  "{l}\\n\\t" \\')
    out[-1] = out[-1][:-2]
    return "\n".join(out) + "\n"


def main():
    txt = "// GENERATED by tools/gen_gemm4w_ktile.py -- do not edit.  See that script for the schedule.\n"
    txt += "#ifndef W4_DIAG\n" + emit("W4_TILE_ASM", body())
    for d in (1, 2, 3):
        txt += f"#elif W4_DIAG == {d}\n" + emit("W4_TILE_ASM", body(d))
    txt += "#else\n" + emit("W4_TILE_ASM", body()) + "#endif\n"
    acc = ", ".join(f'[c{i}{j}] "+a"(acc[{i}][{j}])' for i in range(4) for j in range(4))
    lo = ", ".join(f'[l{r}] "+v"(fl[{r}])' for r in range(8))
    hi = ", ".join(f'[h{r}] "=&v"(fh[{r}])' for r in range(8))
    p = ", ".join(f'[p{r}] "+v"(fp[{r}])' for r in range(16))
    q = ", ".join(f'[q{r}] "=&v"(fq[{r}])' for r in range(16))
    txt += f"\n#define W4_TILE_OUTS {acc}, {lo}, {p}, {hi}, {q}\n"
    ad = ", ".join(f'[{m}{s}{k}] "v"(ad{m.upper()}[{s}][{k}])' for m in "xy" for s in range(2) for k in range(4))
    txt += f"#define W4_TILE_ADDRS {ad}\n"
    open(OUT, "w").write(txt)
    print("wrote", os.path.normpath(OUT), len(body()), "instructions")


if __name__ == "__main__":
    main()
