#!/usr/bin/env python3
"""Generates image-text-retrieval_amd/csrc/gemm_stream_asm.inc: the main body of `gemm_nt_stream_kernel` (csrc/gemm_stream.hip)
as ONE inline-asm statement per epilogue variant, with hand-allocated registers.

What it is for.  `C = act(A B^T + bias)` with a SHORT K (the 256 x 256 projections of the SGRAF graph-reasoning steps, SAEM's
256-wide layers, BERT's 768-wide ones): a 128 x 128 output tile is only K / 32 chunks long, and the tile-at-a-time kernel
(gemm_f32.hip) pays a pipeline fill (first operand chunk: a full L2 / HBM latency) and a drain per tile.  Here a workgroup owns
ONE column tile and STREAMS down its range of row tiles: the operand pipeline never drains at a tile boundary (while the last
chunks of tile t are multiplied, chunks 0 and 1 of tile t+1 are already on their way).

Two accumulator sets.  Tile t accumulates into set t & 1; the OTHER set still holds tile t-1, and its flush (bias, optional
relu, 64 row stores per wave) is slotted behind the MFMAs of tile t's first two chunks, two instruction groups per MFMA.  A
flush issued as one block (the first version of this kernel) cost ~3.4 chunk times per tile: the 8 waves of a compute unit push
their 512 store instructions through one address unit at the same moment and nothing multiplies meanwhile.  The first MFMA
into each accumulator of a tile takes C = 0 as an inline constant, so the sets are never zeroed.

Per 32-wide K chunk (64 x v_mfma_f32_32x32x2_f32 per wave, 4 waves = 2 x 2 of 64 x 64):
    wait the register stage (chunk g+1) | park it in the other LDS buffer | request chunk g+2 | read fragment set F1(g)
    32 MFMAs on F0(g), one memory instruction slotted behind each of the first 24, scalar pointer updates behind the rest
    s_waitcnt lgkmcnt(0); s_barrier
    read F0(g+1) from the other buffer behind the first 8 of the 32 MFMAs on F1(g)
vmcnt counts loads and stores in issue order: the chunk after one that issued S flush stores behind its last operand load waits
with vmcnt(S) -- the loads have landed, the stores need not have been acknowledged.
LDS layout, fragment mapping and the k-ordering are those of gemm_nt_fast_kernel (plane / XOR layout, gemm_f32.hip): results
are bit-identical to that kernel's.  All registers the statement touches are named literally and declared clobbered (see
tools/gen_scan_mainloop.py for why); hipcc keeps its 20 address / bias values in v0..v31.

    python tools/gen_gemm_stream.py [--check]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "image-text-retrieval_amd", "csrc", "gemm_stream_asm.inc")

BM = 128
OPER_BYTES = 8 * BM * 16          # one operand of one chunk: 8 planes x 128 rows x float4
BUF_BYTES = 2 * OPER_BYTES        # A | B

# ---- register map
ACC0 = 32                          # set s, acc[i][j]: 16 registers each: v[32:95] (set 0), v[96:159] (set 1)
FRAG = {(0, 'A'): 160, (0, 'B'): 176, (1, 'A'): 192, (1, 'B'): 208}     # fragment set -> base; [q_ (2)][t (2)] float4
STAGE = 224                        # ra0..3, rb0..3: v[224:255]
V_LO, V_HI = 32, 255
# scalars
S_PA, S_PB, S_PC, S_ROW, S_PCP = 70, 72, 74, 88, 92
S_TSTEP, S_KB, S_LDC4, S_LDC4X5, S_NT, S_NK2, S_LK, S_NKM1, S_T0, S_T1, S_LT, S_T2, S_CNT = 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 90
S_C707, S_ABSMASK = 94, 95           # gelu epilogue: 1 / sqrt(2), 0x7fffffff
S_LO, S_HI = 70, 95
NF = 2                             # flush groups slotted behind one MFMA (relu / none); gelu: NF_GELU single instructions
NF_GELU = 6
# gelu_erf of csrc/itr_common.h, instruction for instruction (Q coefficients t^7 .. t^0): the streamed tile must equal the tile kernel's
GELU_Q = ["0x3803def8", "0xb9a1e718", "0x3a875201", "0x3a0cbf85", "0xbca06e7e", "0x3dd2c783", "0x3f22f812", "0x3f906ec4"]


def v4(b):
    return "v[%d:%d]" % (b, b + 3)


def acc_base(s, i, j):
    return ACC0 + 64 * s + (i * 2 + j) * 16


def acc(s, i, j):
    b = acc_base(s, i, j)
    return "v[%d:%d]" % (b, b + 15)


def lstore(buf):
    o = buf * BUF_BYTES
    ins = ["ds_write_b128 %%[ls0], %s offset:%d" % (v4(STAGE + 4 * s), o + s * 32 * 16) for s in range(4)]
    ins += ["ds_write_b128 %%[ls0], %s offset:%d" % (v4(STAGE + 16 + 4 * s), o + OPER_BYTES + s * 32 * 16) for s in range(4)]
    return ins


def gload():
    """8 loads of the chunk the running pointers address, and the pointer update as groups of scalar instructions (a group is
    never split: its members talk through SCC)."""
    ins = []
    for s in range(4):     # A and B passes alternate like the C++ kernel does
        ins.append("global_load_dwordx4 %s, %%[oa%d], s[%d:%d]" % (v4(STAGE + 4 * s), s, S_PA, S_PA + 1))
        ins.append("global_load_dwordx4 %s, %%[ob%d], s[%d:%d]" % (v4(STAGE + 16 + 4 * s), s, S_PB, S_PB + 1))
    # advance the load stream by one chunk; at the end of a row tile step to the next one (or stay on the last one)
    adv = [["s_cmp_eq_u32 s%d, s%d" % (S_LK, S_NKM1),                      # SCC = this was the last chunk of its row tile
            "s_cselect_b32 s%d, s%d, 0" % (S_T1, S_KB),                    # rewind by K bytes ...
            "s_cselect_b32 s%d, s%d, 0" % (S_T2, S_TSTEP),                 # ... and step one row tile down
            "s_cselect_b32 s%d, -1, s%d" % (S_LK, S_LK)],
           ["s_add_u32 s%d, s%d, 1" % (S_LK, S_LK)],
           ["s_cmp_gt_u32 s%d, 1" % S_LT,                                  # is there another row tile to fetch?
            "s_cselect_b32 s%d, s%d, 0" % (S_T2, S_T2)],
           ["s_cmp_lg_u32 s%d, 0" % S_T2,
            "s_cselect_b32 s%d, 1, 0" % S_T0],
           ["s_sub_u32 s%d, s%d, s%d" % (S_LT, S_LT, S_T0)],
           ["s_add_u32 s%d, s%d, 128" % (S_PA, S_PA), "s_addc_u32 s%d, s%d, 0" % (S_PA + 1, S_PA + 1)],
           ["s_sub_u32 s%d, s%d, s%d" % (S_PA, S_PA, S_T1), "s_subb_u32 s%d, s%d, 0" % (S_PA + 1, S_PA + 1)],
           ["s_add_u32 s%d, s%d, s%d" % (S_PA, S_PA, S_T2), "s_addc_u32 s%d, s%d, 0" % (S_PA + 1, S_PA + 1)],
           ["s_add_u32 s%d, s%d, 128" % (S_PB, S_PB), "s_addc_u32 s%d, s%d, 0" % (S_PB + 1, S_PB + 1)],
           ["s_sub_u32 s%d, s%d, s%d" % (S_PB, S_PB, S_T1), "s_subb_u32 s%d, s%d, 0" % (S_PB + 1, S_PB + 1)]]
    return ins, adv


def fread(fset, buf):
    """8 x ds_read_b128: fragment set fset (0: k-planes 0..3, 1: planes 4..7) of the chunk in LDS buffer buf."""
    o = buf * BUF_BYTES
    ins = []
    for q_ in range(2):
        q = 2 * fset + q_
        for t in range(2):
            ins.append("ds_read_b128 %s, %%[fa%d] offset:%d" % (v4(FRAG[(fset, 'A')] + (q_ * 2 + t) * 4), q, o + t * 32 * 16))
            ins.append("ds_read_b128 %s, %%[fb%d] offset:%d" % (v4(FRAG[(fset, 'B')] + (q_ * 2 + t) * 4), q, o + t * 32 * 16))
    return ins


def mfmas(aset, fset, first=False):
    """32 MFMAs of one fragment set, component-major: consecutive instructions never touch the same accumulator.
    first: the four that open the tile take C = 0."""
    out = []
    for q_ in range(2):
        for c in range(4):
            for i in range(2):
                for j in range(2):
                    a = FRAG[(fset, 'A')] + (q_ * 2 + i) * 4 + c
                    b = FRAG[(fset, 'B')] + (q_ * 2 + j) * 4 + c
                    src = "0" if (first and q_ == 0 and c == 0) else acc(aset, i, j)
                    out.append("v_mfma_f32_32x32x2_f32 %s, v%d, v%d, %s" % (acc(aset, i, j), a, b, src))
    return out


def gelu_insts(reg):
    """x = v[reg] (bias already added) -> gelu(x) in place; temporaries %[t0] (t), %[t1] (q, then erf), %[t2] (x / 2).
    One instruction per group.  The independent v_mul between v_exp_f32 and its consumer covers the trans -> VALU wait state."""
    ins = ["v_mul_f32_e64 %%[t0], |v%d|, s%d" % (reg, S_C707),
           "v_min_f32 %[t0], 4.0, %[t0]",
           "v_mov_b32 %%[t1], %s" % GELU_Q[0]]
    for c in GELU_Q[1:]:
        ins.append("v_fmaak_f32 %%[t1], %%[t1], %%[t0], %s" % c)
    ins += ["v_mul_f32 %[t1], %[t1], %[t0]",
            "v_mul_f32 %[t1], 0xbfb8aa3b, %[t1]",
            "v_exp_f32 %[t1], %[t1]",
            "v_mul_f32 %%[t2], 0.5, v%d" % reg,
            "v_sub_f32 %[t1], 1.0, %[t1]",
            "v_bfi_b32 %%[t1], s%d, %%[t1], v%d" % (S_ABSMASK, reg),
            "v_fma_f32 v%d, %%[t2], %%[t1], %%[t2]" % reg]
    return [[x] for x in ins]


def flush_groups(fs, act):
    """Flush of accumulator set fs to the tile at s[S_ROW], as groups: + bias, [relu | gelu], store, row by row.  C/D layout of
    v_mfma_f32_32x32x2_f32: register r of lane (fi = lane & 31, fg = lane >> 5) is row (r & 3) + 8 (r >> 2) + 4 fg, column fi.
    Rows of acc[0][*] first: the MFMAs that finished the tile ended on acc[1][1].  act: 0 none, 1 relu, 4 gelu."""
    gs = []
    for idx in range(32):
        i, r = idx // 16, idx % 16
        for j in range(2):
            reg = acc_base(fs, i, j) + r
            gs.append(["v_add_f32 v%d, v%d, %%[bias%d]" % (reg, reg, j)])
            if act == 1:
                gs.append(["v_max_f32 v%d, v%d, 0" % (reg, reg)])
            elif act == 4:
                gs += gelu_insts(reg)
            gs.append(["global_store_dword %%[voffc], v%d, s[%d:%d] offset:%d" % (reg, S_ROW, S_ROW + 1, j * 128)])
        step = S_LDC4X5 if idx % 4 == 3 else S_LDC4
        gs.append(["s_add_u32 s%d, s%d, s%d" % (S_ROW, S_ROW, step), "s_addc_u32 s%d, s%d, 0" % (S_ROW + 1, S_ROW + 1)])
    return gs


def stores_behind_last_load(ins):
    last = max(n for n, x in enumerate(ins) if x.startswith("global_load"))
    return sum(1 for x in ins[last:] if x.startswith("global_store"))


def slot(mf, mem, fl=(), skip=0, nf=NF):
    """Behind MFMA number n: mem[n] (an instruction or a group), then nf groups of fl (from MFMA `skip` on).  Returns the
    instruction list and what is left of fl."""
    out, fl = [], list(fl)
    for n, x in enumerate(mf):
        out.append(x)
        if n < len(mem):
            out += mem[n] if isinstance(mem[n], list) else [mem[n]]
        if n >= skip:
            for g in fl[:nf]:
                out += g
            fl = fl[nf:]
    for m in mem[len(mf):]:
        out += m if isinstance(m, list) else [m]
    return out, fl


def chunk(aset, cur, nxt, vmn=0, first=False, fl=(), nf=NF):
    """One chunk into accumulator set aset.  vmn: stores the previous chunk issued behind the operand loads this one waits for.
    Returns (instructions, flush groups left over, stores issued behind this chunk's last operand load)."""
    ins = ["s_waitcnt vmcnt(%d) lgkmcnt(0)" % vmn]   # the stage (chunk g+1) landed; F0(g), read during the previous half, landed
    ld, adv = gload()
    body, fl = slot(mfmas(aset, 0, first), lstore(nxt) + ld + fread(1, cur) + adv, fl, skip=2 if first else 0, nf=nf)
    ins += body
    ins += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    body, fl = slot(mfmas(aset, 1), fread(0, nxt), fl, nf=nf)
    ins += body
    return ins, fl, stores_behind_last_load(ins)


def flush_block(fs, act):
    """The last tile's flush, on its own."""
    ins = ["s_nop 15", "s_nop 7", "s_mov_b64 s[%d:%d], s[%d:%d]" % (S_ROW, S_ROW + 1, S_PCP, S_PCP + 1)]   # 16-pass MFMA results -> VALU
    for g in flush_groups(fs, act):
        ins += g
    return ins


def tile(aset, act, L):
    """One row tile into set aset while set 1 - aset (the previous tile) is flushed.  >= 4 chunks (K >= 128).  The flush is over
    within the first two chunks (none / relu: 2 groups per MFMA) or the first four (gelu: NF_GELU instructions per MFMA)."""
    t = "t%d" % aset
    ins = [L(t) + ":", "s_mov_b64 s[%d:%d], s[%d:%d]" % (S_ROW, S_ROW + 1, S_PCP, S_PCP + 1)]
    fl = flush_groups(1 - aset, act)
    nf = NF_GELU if act == 4 else NF
    c, fl, s0 = chunk(aset, 0, 1, 0, True, fl, nf)
    ins += c
    c, fl, s1 = chunk(aset, 1, 0, s0, False, fl, nf)
    ins += c
    assert s0 <= 63 and s1 <= 63, (s0, s1)
    c, fl, s2 = chunk(aset, 0, 1, s1, False, fl, nf)
    ins += c
    c, fl, s3 = chunk(aset, 1, 0, s2, False, fl, nf)
    ins += c
    assert not fl and s2 <= 63 and s3 <= 63, (len(fl), s2, s3)
    assert act == 4 or (s2 == 0 and s3 == 0)             # (none / relu: chunks 2 and 3 are the plain ones of before)
    ins += ["s_sub_u32 s%d, s%d, 2" % (S_CNT, S_NK2), "s_cmp_eq_u32 s%d, 0" % S_CNT, "s_cbranch_scc1 " + L(t + "_end"), L(t + "_chunk") + ":"]
    ins += chunk(aset, 0, 1)[0] + chunk(aset, 1, 0)[0]
    ins += ["s_sub_u32 s%d, s%d, 1" % (S_CNT, S_CNT), "s_cmp_lg_u32 s%d, 0" % S_CNT, "s_cbranch_scc1 " + L(t + "_chunk"), L(t + "_end") + ":"]
    # this tile becomes the one to flush; the output pointer moves one row tile down
    ins += ["s_mov_b64 s[%d:%d], s[%d:%d]" % (S_PCP, S_PCP + 1, S_PC, S_PC + 1),
            "s_lshl_b32 s%d, s%d, 7" % (S_T0, S_LDC4), "s_add_u32 s%d, s%d, s%d" % (S_PC, S_PC, S_T0), "s_addc_u32 s%d, s%d, 0" % (S_PC + 1, S_PC + 1),
            "s_sub_u32 s%d, s%d, 1" % (S_NT, S_NT), "s_cmp_eq_u32 s%d, 0" % S_NT, "s_cbranch_scc1 " + L("last%d" % aset)]
    return ins


def program(act):
    L = lambda name: ".Lgs%d_%s_%%=" % (act, name)
    ins = ["s_mov_b64 s[%d:%d], %%[pa]" % (S_PA, S_PA + 1), "s_mov_b64 s[%d:%d], %%[pb]" % (S_PB, S_PB + 1),
           "s_mov_b64 s[%d:%d], %%[pc]" % (S_PC, S_PC + 1), "s_mov_b64 s[%d:%d], %%[pc]" % (S_PCP, S_PCP + 1),
           "s_mov_b32 s%d, %%[tstep]" % S_TSTEP, "s_mov_b32 s%d, %%[kbytes]" % S_KB,
           "s_mov_b32 s%d, %%[ldc4]" % S_LDC4, "s_mul_i32 s%d, s%d, 5" % (S_LDC4X5, S_LDC4), "s_mov_b32 s%d, %%[ntile]" % S_NT,
           "s_mov_b32 s%d, %%[ntile]" % S_LT, "s_mov_b32 s%d, %%[nk2]" % S_NK2, "s_lshl_b32 s%d, s%d, 1" % (S_NKM1, S_NK2),
           "s_sub_u32 s%d, s%d, 1" % (S_NKM1, S_NKM1), "s_mov_b32 s%d, 0" % S_LK,
           "s_mov_b32 s%d, 0x3f3504f3" % S_C707, "s_mov_b32 s%d, 0x7fffffff" % S_ABSMASK]
    # The first tile "flushes" set 1 to its own output tile (bias / relu(bias) rows, overwritten by its real flush later).
    ins += ["v_mov_b32 v%d, 0" % r for r in range(ACC0 + 64, ACC0 + 128)]
    # prologue: chunk 0 -> LDS buffer 0, chunk 1 -> stage
    ld, adv = gload()
    ins += ld + sum(adv, []) + ["s_waitcnt vmcnt(0)"] + lstore(0)
    ld, adv = gload()
    ins += ld + sum(adv, []) + ["s_waitcnt lgkmcnt(0)", "s_barrier"] + fread(0, 0)
    ins += tile(0, act, L) + tile(1, act, L) + ["s_branch " + L("t0")]
    ins += [L("last0") + ":"] + flush_block(0, act) + ["s_branch " + L("done")]
    ins += [L("last1") + ":"] + flush_block(1, act)
    ins += [L("done") + ":", "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]      # the tail prefetch (re-read of the last tile) and the stores are done
    return ins


def render_one(act):
    ins = program(act)
    lines = ["    asm volatile("]
    for x in ins:
        lines.append('        "%s\\n\\t"' % x)
    lines.append('        : [t0] "=&v"(gt0), [t1] "=&v"(gt1), [t2] "=&v"(gt2)' if act == 4 else "        :")
    lines.append('        : [oa0] "v"(oa0), [oa1] "v"(oa1), [oa2] "v"(oa2), [oa3] "v"(oa3), [ob0] "v"(ob0), [ob1] "v"(ob1), [ob2] "v"(ob2), [ob3] "v"(ob3),')
    lines.append('          [ls0] "v"(ls0), [fa0] "v"(fa0), [fa1] "v"(fa1), [fa2] "v"(fa2), [fa3] "v"(fa3), [fb0] "v"(fb0), [fb1] "v"(fb1), [fb2] "v"(fb2), [fb3] "v"(fb3),')
    lines.append('          [voffc] "v"(voffc), [bias0] "v"(bias0), [bias1] "v"(bias1),')
    lines.append('          [pa] "s"(pa), [pb] "s"(pb), [pc] "s"(pc), [tstep] "s"(tstep), [kbytes] "s"(kbytes), [ldc4] "s"(ldc4), [ntile] "s"(ntile), [nk2] "s"(nk2)')
    clob = ['"memory"', '"scc"'] + ['"s%d"' % s for s in range(S_LO, S_HI + 1)] + ['"v%d"' % r for r in range(V_LO, V_HI + 1)]
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


def render():
    out = ["// GENERATED by tools/gen_gemm_stream.py -- do not edit; regenerate and commit (tests/test_isa_audit.py checks it is current).",
           "// Body of gemm_nt_stream_kernel<ACT> (0 none, 1 relu, 4 gelu): prologue + tile loop (two accumulator sets; the previous tile is flushed behind the",
           "// MFMAs of the current one) with hand-allocated registers.",
           "    if constexpr (ACT == 1) {", render_one(1).rstrip("\n"), "    } else if constexpr (ACT == 4) {", "    float gt0, gt1, gt2;      // temporaries of the gelu flush",
           render_one(4).rstrip("\n"), "    } else {", render_one(0).rstrip("\n"), "    }"]
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    open(OUT, "w").write(text)
    print("wrote %s: %d lines" % (OUT, text.count("\n")))
