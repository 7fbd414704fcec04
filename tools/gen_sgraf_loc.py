#!/usr/bin/env python3
"""Generates image-text-retrieval_amd/csrc/sgraf_loc_asm.inc: the D loop of `sgraf_loc_kernel` (csrc/sgraf_loc.hip) as ONE
inline-asm statement with hand-allocated registers (see tools/gen_scan_mainloop.py for why such loops are not left to hipcc).

Per 32-wide slice k of D a wave runs 64 x v_mfma_f32_32x32x2_f32 (stage 2: acc[64 x 64] += a_k W_k^T from LDS) and PRODUCES
slice k+1 of the operand: 18 x v_mfma_f32_16x16x4_f32 (stage 1: ctx = P' V), (ctx - E)^2 on the vector ALU, 8 x ds_write_b32 in
the A-operand layout, plus its share of W_loc's slice (8 x ds_write_b128).  The C++ loop did stage 1 after stage 2 and met a
barrier before the next slice's first fragment read: ~1 100 of 5 800 cycles per slice were exposed.  Here

    top     s_waitcnt vmcnt(0) lgkmcnt(0)         the register stage (V, E, W of slice k+1) and fragment set F0(k) landed
    half 1  32 big MFMAs on F0(k); behind them: the stage-1 MFMAs of slice k+1 (two after each of the first nine), the W stores,
            the refill of every stage register right behind its last reader (slice k+2), the squared difference and its stores,
            the reads of fragment set F1(k), the scalar pointer updates
            s_waitcnt lgkmcnt(0); s_barrier       slice k+1 is in the other LDS buffer, every wave is done with F1(k)'s buffer
    half 2  32 big MFMAs on F1(k); behind the first 8: the reads of F0(k+1)

so the matrix core never waits for a barrier or an LDS round trip.  The last iteration produces a dummy slice (the pointers stop
at the last one) into the idle buffer.  The accumulators leave the statement in v[64:127] (physical-register outputs).

    python tools/gen_sgraf_loc.py [--check]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "image-text-retrieval_amd", "csrc", "sgraf_loc_asm.inc")

A_BUF, W_OFF, W_BUF = 8192, 16384, 32768      # LocSmem: a[2][8][64] float4, then w[2][8][256] float4

# ---- register map
C0, ACC0, F0, F1, RW, VB, ZZ = 48, 64, 128, 160, 192, 224, 242
CLOB_V = list(range(48, 56)) + list(range(128, 250))
SV, SZ, SW, S_ADV, S_REM, S_CNT, S_T = 40, 58, 66, 82, 83, 84, 85
S_LO, S_HI = 40, 85


def v4(b):
    return "v[%d:%d]" % (b, b + 3)


def acc(i, j):
    b = ACC0 + (i * 2 + j) * 16
    return "v[%d:%d]" % (b, b + 15)


def sp(b):
    return "s[%d:%d]" % (b, b + 1)


def load_v(nt, q):
    return "global_load_dword v%d, %%[voffv], %s%s" % (VB + 9 * nt + q, sp(SV + 2 * q), " offset:64" if nt else "")


def load_z(nt, j):
    return "global_load_dword v%d, %%[voffz], %s%s" % (ZZ + 4 * nt + j, sp(SZ + 2 * j), " offset:64" if nt else "")


def load_w(s):
    return "global_load_dwordx4 %s, %%[voffw], %s" % (v4(RW + 4 * s), sp(SW + 2 * s))


def store_w(s, buf):
    return "ds_write_b128 %%[wst], %s offset:%d" % (v4(RW + 4 * s), buf * W_BUF + s * 512)


def advance():
    """Step the 21 row pointers one slice on, or leave them on the last slice.  Groups talk through SCC and are never split."""
    gs = [["s_cmp_lg_u32 s%d, 0" % S_REM, "s_cselect_b32 s%d, 128, 0" % S_ADV, "s_cselect_b32 s%d, 1, 0" % S_T],
          ["s_sub_u32 s%d, s%d, s%d" % (S_REM, S_REM, S_T)]]
    for b in [SV + 2 * q for q in range(9)] + [SZ + 2 * j for j in range(4)] + [SW + 2 * s for s in range(8)]:
        gs.append(["s_add_u32 s%d, s%d, s%d" % (b, b, S_ADV), "s_addc_u32 s%d, s%d, 0" % (b + 1, b + 1)])
    return gs


def small(nt, q):
    c = v4(C0 + 4 * nt)
    return "v_mfma_f32_16x16x4_f32 %s, %%[pa%d], v%d, %s" % (c, q, VB + 9 * nt + q, "0" if q == 0 else c)


def sqdiff(nt, j, buf):
    c = C0 + 4 * nt + j
    return ["v_sub_f32 v%d, v%d, v%d" % (c, c, ZZ + 4 * nt + j), "v_mul_f32 v%d, v%d, v%d" % (c, c, c),
            "ds_write_b32 %%[ast%d%d], v%d offset:%d" % (nt, j, c, buf * A_BUF)]


def fread(fset, buf):
    base = F0 if fset == 0 else F1
    ins = []
    for q_ in range(2):
        q = 2 * fset + q_
        for t in range(2):
            ins.append("ds_read_b128 %s, %%[fa%d] offset:%d" % (v4(base + q_ * 16 + t * 4), q, buf * A_BUF + t * 512))
            ins.append("ds_read_b128 %s, %%[fb%d] offset:%d" % (v4(base + q_ * 16 + 8 + t * 4), q, buf * W_BUF + t * 512))
    return ins


def bigs(fset):
    base = F0 if fset == 0 else F1
    out = []
    for q_ in range(2):
        for c in range(4):
            for i in range(2):
                for j in range(2):
                    # operands swapped (W fragment as A, node fragment as B): the accumulator holds the tile TRANSPOSED -- a lane owns
                    # ONE node row and four runs of four consecutive features, which the epilogue stores as 16-byte rows
                    out.append("v_mfma_f32_32x32x2_f32 %s, v%d, v%d, %s" % (acc(i, j), base + q_ * 16 + 8 + j * 4 + c, base + q_ * 16 + i * 4 + c, acc(i, j)))
    return out


def iteration(cur):
    nxt = cur ^ 1
    slots = [[] for _ in range(32)]
    for m in range(9):                                   # stage 1 of slice k+1: both chains advance behind big MFMA m
        slots[m] += [small(0, m), small(1, m)]
    for s in range(8):
        slots[s].append(store_w(s, nxt))                 # W of slice k+1 -> LDS
    for m in range(9):
        slots[m + 1] += [load_v(0, m), load_v(1, m)]     # refill behind the reader (slice k+2)
    for s in range(8):
        slots[8 + s].append(load_w(s))
    n = 0
    for nt in range(2):
        for j in range(4):
            slots[11 + n] += sqdiff(nt, j, nxt)          # >= 2 big MFMAs behind the last stage-1 MFMA
            slots[12 + n].append(load_z(nt, j))
            n += 1
    for n, x in enumerate(fread(1, cur)):
        slots[19 + n].append(x)
    adv = advance()
    per = (len(adv) + 11) // 12
    for n in range(12):
        for g in adv[n * per:(n + 1) * per]:
            slots[20 + n] += g
    ins = ["s_waitcnt vmcnt(0) lgkmcnt(0)"]
    for b, s in zip(bigs(0), slots):
        ins += [b] + s
    ins += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    rd = fread(0, nxt)
    for n, b in enumerate(bigs(1)):
        ins.append(b)
        if n < len(rd):
            ins.append(rd[n])
    return ins


def program():
    L = lambda name: ".Lloc_%s_%%=" % name
    ins = ["s_mov_b64 %s, %%[vbase]" % sp(SV), "s_lshl_b32 s%d, %%[rowb], 2" % S_T]
    for q in range(1, 9):
        ins += ["s_add_u32 s%d, s%d, s%d" % (SV + 2 * q, SV + 2 * q - 2, S_T), "s_addc_u32 s%d, s%d, 0" % (SV + 2 * q + 1, SV + 2 * q - 1)]
    ins += ["s_mov_b64 %s, %%[zbase]" % sp(SZ)]
    for j in range(1, 4):
        ins += ["s_add_u32 s%d, s%d, %%[rowb]" % (SZ + 2 * j, SZ + 2 * j - 2), "s_addc_u32 s%d, s%d, 0" % (SZ + 2 * j + 1, SZ + 2 * j - 1)]
    ins += ["s_mov_b64 %s, %%[wbase]" % sp(SW), "s_lshl_b32 s%d, %%[rowb], 5" % S_T]
    for s in range(1, 8):
        ins += ["s_add_u32 s%d, s%d, s%d" % (SW + 2 * s, SW + 2 * s - 2, S_T), "s_addc_u32 s%d, s%d, 0" % (SW + 2 * s + 1, SW + 2 * s - 1)]
    ins += ["s_sub_u32 s%d, %%[nk], 1" % S_REM, "s_mov_b32 s%d, %%[nk]" % S_CNT]
    loads = [load_v(nt, q) for q in range(9) for nt in range(2)] + [load_z(nt, j) for nt in range(2) for j in range(4)] + [load_w(s) for s in range(8)]
    # the accumulators start at the bias (transposed tile: element r of acc(i, j) is feature wave*64 + 32 j + 8 (r >> 2) + 4 fg + (r & 3),
    # %[bptr] = &bias[wave*64 + 4 fg]): four 16-byte loads per j into acc(0, j), copied to acc(1, j) below -- 64 adds fewer in the epilogue
    bias_loads = ["global_load_dwordx4 v[%d:%d], %%[bptr], off offset:%d" % (ACC0 + j * 16 + 4 * b, ACC0 + j * 16 + 4 * b + 3, (32 * j + 8 * b) * 4)
                  for j in range(2) for b in range(4)]
    # prologue: slice 0 -> LDS buffer 0, slice 1 -> register stage
    ins += loads + bias_loads + sum(advance(), []) + ["s_waitcnt vmcnt(0)"]
    ins += [small(nt, q) for q in range(9) for nt in range(2)] + ["s_nop 15"]
    for nt in range(2):
        for j in range(4):
            ins += sqdiff(nt, j, 0)
    ins += [store_w(s, 0) for s in range(8)]
    ins += loads + sum(advance(), [])
    ins += ["v_mov_b32 v%d, v%d" % (ACC0 + 32 + r, ACC0 + r) for r in range(32)]          # acc(1, j) = acc(0, j) = bias
    ins += ["s_waitcnt lgkmcnt(0)", "s_barrier"] + fread(0, 0)
    ins += [L("loop") + ":"] + iteration(0)
    ins += ["s_sub_u32 s%d, s%d, 1" % (S_CNT, S_CNT), "s_cmp_eq_u32 s%d, 0" % S_CNT, "s_cbranch_scc1 " + L("done")]
    ins += iteration(1)
    ins += ["s_sub_u32 s%d, s%d, 1" % (S_CNT, S_CNT), "s_cmp_lg_u32 s%d, 0" % S_CNT, "s_cbranch_scc1 " + L("loop")]
    ins += [L("done") + ":", "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15", "s_nop 7"]     # nothing in flight; 16-pass MFMA results readable
    return ins


def render():
    ins = program()
    lines = ["// GENERATED by tools/gen_sgraf_loc.py -- do not edit; regenerate and commit (tests/test_isa_audit.py checks it is current).",
             "// D loop of sgraf_loc_kernel: prologue + slice loop with hand-allocated registers; accumulators returned in v[64:127].",
             "    asm volatile("]
    for x in ins:
        lines.append('        "%s\\n\\t"' % x)
    lines.append('        : "=&{v[64:79]}"(acc[0][0]), "=&{v[80:95]}"(acc[0][1]), "=&{v[96:111]}"(acc[1][0]), "=&{v[112:127]}"(acc[1][1])')
    lines.append("        : " + ", ".join('[pa%d] "v"(pa[%d])' % (q, q) for q in range(9)) + ",")
    lines.append('          [voffv] "v"(voff_v), [voffz] "v"(voff_z), [voffw] "v"(voff_w), [wst] "v"(wst),')
    lines.append("          " + ", ".join('[ast%d%d] "v"(ast[%d][%d])' % (nt, j, nt, j) for nt in range(2) for j in range(4)) + ",")
    lines.append("          " + ", ".join('[fa%d] "v"(fa[%d])' % (q, q) for q in range(4)) + ", " + ", ".join('[fb%d] "v"(fb[%d])' % (q, q) for q in range(4)) + ",")
    lines.append('          [vbase] "s"(vbase), [zbase] "s"(zbase), [wbase] "s"(wbase), [rowb] "s"(rowb), [nk] "s"(nk), [bptr] "v"(bias_lane)')
    clob = ['"memory"', '"scc"'] + ['"s%d"' % s for s in range(S_LO, S_HI + 1)] + ['"v%d"' % r for r in CLOB_V]
    rows, row = [], []
    for c in clob:
        row.append(c)
        if len(row) == 16:
            rows.append(", ".join(row))
            row = []
    if row:
        rows.append(", ".join(row))
    lines.append("        : " + (",\n          ".join(rows)) + ");")
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    open(OUT, "w").write(text)
    print("wrote %s: %d lines, %d MFMAs" % (OUT, text.count("\n"), text.count("v_mfma")))
