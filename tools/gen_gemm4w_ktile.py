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


# ------------------------------------------------------------------------------------------------------------------------------------
# TN (weight gradient: C = A^T B, A [K, M], B [K, N], k along the rows of both operands).  Same X / Y schedule; what differs:
#   * a staged k-tile = eight 8-KiB sub-images [64 k][64 columns] (128-B rows: A columns wr*128 + {0..63 | 64..127}, B likewise; the
#     16-B chunk index of k-row r is XORed with 4 ((r >> 1) & 1)); wave w stages A sub-image w and B sub-image w: 8 pieces of 8 k-rows;
#     the pieces' k-rows beyond the slice's K range read a zero block instead (per-piece scalar select)
#   * a fragment = two ds_read_b64_tr_b16 (k rows +0 and +4) -> the fragment registers are PHYSICAL (v64..v255: an asm operand cannot
#     name half of a register quadruple); the first fragments are read inside the asm
#   * optional bias-gradient column sums: v_dot2_f32_bf16 of every A fragment against ones (cs0..3 = the wave's four row blocks)
# Read addresses: a{E|O}{slot} / b{E|O}{slot} = the lane's address in 32-column block 0 / 1 of the wave's A / B sub-image 0 of the slot
# (sub-image 1 = + 8192; k-step ks = + ks * 2048; second half = + 512).
def FR(kind, idx):  # first register of fragment idx of set l | h | p | q
    return {"l": 64, "h": 96, "p": 128, "q": 192}[kind] + 4 * idx


def tn_frag_reads(kind, idx, addr, off):
    r = FR(kind, idx)
    return [f"ds_read_b64_tr_b16 v[{r}:{r + 1}], %[{addr}] offset:{off}", f"ds_read_b64_tr_b16 v[{r + 2}:{r + 3}], %[{addr}] offset:{off + 512}"]


def tn_piece_head(L, which):
    """valid / zero-block selection of a piece, at the HEAD of its group: two MFMAs ahead of the load that consumes the selected base
    and offset (an SALU-written SGPR needs wait states before a VMEM instruction reads it, and the chain s_cmp -> s_cselect ->
    v_cndmask in front of the load stalled the in-order issue)"""
    L += [f"s_cmp_gt_i32 %[kr{which}], 0", f"s_cselect_b64 %[sb], %[mat{which}], %[zb]", "s_cselect_b64 vcc, -1, 0",
          f"v_cndmask_b32 %[vt], %[zoff], %[pe{which}], vcc"]


def tn_piece(L, which, n, slot):
    L += [f"global_load_lds_dwordx4 %[vt], %[sb]", f"v_add_u32 %[pe{which}], %[step{which}], %[pe{which}]",
          f"s_sub_u32 %[kr{which}], %[kr{which}], 8"]


def tn_ktile(L, slot, bc, bn, csum, diag):
    o = 1 - slot
    L += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    # ---- X: rows 0..63 (A-lo x B); fetch A-hi of this k-tile; stage B of the k-tile after next
    reads = []
    for ks in range(4):
        for b in range(2):
            reads += tn_frag_reads("h", ks * 2 + b, f"a{'EO'[b]}{slot}", 8192 + ks * 2048)
    dots = [(f"cs{j}", FR("l", ks * 2 + j) + w) for ks in range(4) for j in range(2) for w in range(4)]
    for g in range(16):
        ks, q = g >> 2, g & 3
        j, i0 = q >> 1, (q & 1) * 2
        if g in DMA_GROUPS and diag != 3:
            L.append(f"s_mov_b32 m0, %[db{slot}]" if g == 0 else "s_add_u32 m0, m0, 0x400")
            tn_piece_head(L, "b")
        for i in (i0, i0 + 1):
            rb, ra = FR(bc, ks * 4 + i), FR("l", ks * 2 + j)
            L.append(f"v_mfma_f32_32x32x16_bf16 %[c{i}{j}], v[{rb}:{rb + 3}], v[{ra}:{ra + 3}], %[c{i}{j}]")
            if i == i0:
                L.append(reads[g])
                if csum:
                    for d, r in dots[2 * g:2 * g + 2]:
                        L.append(f"v_dot2_f32_bf16 %[{d}], v{r}, %[ones], %[{d}]")
        if g in DMA_GROUPS and diag != 3:
            tn_piece(L, "b", DMA_GROUPS.index(g), slot)
    L += ["s_waitcnt vmcnt(8) lgkmcnt(0)", "s_barrier"]
    # ---- Y: rows 64..127 (A-hi x B); fetch A-lo + B of the next k-tile (other slot); stage A of the k-tile after next
    reads = []
    for ks in range(4):
        for b in range(2):
            reads += tn_frag_reads("l", ks * 2 + b, f"a{'EO'[b]}{o}", ks * 2048)
        for i in range(4):
            reads += tn_frag_reads(bn, ks * 4 + i, f"b{'EO'[i & 1]}{o}", (i >> 1) * 8192 + ks * 2048)
    dots = [(f"cs{2 + j}", FR("h", ks * 2 + j) + w) for ks in range(4) for j in range(2) for w in range(4)]
    for g in range(16):
        ks, q = g >> 2, g & 3
        j, i0 = q >> 1, (q & 1) * 2
        if g in DMA_GROUPS and diag != 3:
            L.append(f"s_mov_b32 m0, %[da{slot}]" if g == 0 else "s_add_u32 m0, m0, 0x400")
            tn_piece_head(L, "a")
        for i in (i0, i0 + 1):
            rb, ra = FR(bc, ks * 4 + i), FR("h", ks * 2 + j)
            L.append(f"v_mfma_f32_32x32x16_bf16 %[c{i}{2 + j}], v[{rb}:{rb + 3}], v[{ra}:{ra + 3}], %[c{i}{2 + j}]")
            L += reads[3 * g + (0 if i == i0 else 2):3 * g + (2 if i == i0 else 3)]
            if csum and i == i0:
                for d, r in dots[2 * g:2 * g + 2]:
                    L.append(f"v_dot2_f32_bf16 %[{d}], v{r}, %[ones], %[{d}]")
        if g in DMA_GROUPS and diag != 3:
            tn_piece(L, "a", DMA_GROUPS.index(g), slot)


def tn_body(csum, diag=0):
    L = ["s_mov_b32 %[sm], m0"]
    for ks in range(4):  # the first fragments: A-lo and B (set p) of k-tile 0 in ring slot 0
        for b in range(2):
            L += tn_frag_reads("l", ks * 2 + b, f"a{'EO'[b]}0", ks * 2048)
        for i in range(4):
            L += tn_frag_reads("p", ks * 4 + i, f"b{'EO'[i & 1]}0", (i >> 1) * 8192 + ks * 2048)
    L.append("1:")
    tn_ktile(L, 0, "p", "q", csum, diag)
    tn_ktile(L, 1, "q", "p", csum, diag)
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
    # ---- TN
    txt = "// GENERATED by tools/gen_gemm4w_ktile.py -- do not edit.  See that script for the schedule.\n"
    txt += emit("W4T_TILE_ASM", tn_body(False)) + "\n" + emit("W4T_TILE_ASM_CSUM", tn_body(True))
    # both loops behind ONE scalar branch in ONE asm statement (%[docs] != 0: with column sums).  As two statements in the two arms of a
    # C++ branch, hipcc merged the 256 accumulator registers of the arms through scratch (64 spilled VGPRs, round-5 review)
    both = ["s_cmp_eq_u32 %[docs], 0", "s_cbranch_scc1 8f"] + tn_body(True) + ["s_branch 9f", "8:"] + tn_body(False) + ["9:"]
    txt += "\n" + emit("W4T_TILE_ASM_BOTH", both)
    txt += f"\n#define W4T_TILE_ACC {acc}\n"
    txt += "#define W4T_TILE_CLOBBERS " + ", ".join(f'"v{r}"' for r in range(64, 256)) + "\n"
    open(OUT.replace("gemm4w_ktile", "gemm4w_tn_ktile"), "w").write(txt)
    print("wrote gemm4w_tn_ktile.inc", len(tn_body(True)), "instructions (with column sums)")


if __name__ == "__main__":
    main()
