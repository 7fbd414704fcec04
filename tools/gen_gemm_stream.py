#!/usr/bin/env python3
"""Generates image-text-retrieval_amd/csrc/gemm_stream_asm.inc: the main body of `gemm_nt_stream_kernel` (csrc/gemm_stream.hip)
as ONE inline-asm statement per epilogue variant, with hand-allocated registers.

What it is for.  `C = act(A B^T + bias)` with a SHORT K (the 256 x 256 projections of the SGRAF graph-reasoning steps, SAEM's
256-wide layers): a 128 x 128 output tile is only K / 32 = 8 chunks long, and the tile-at-a-time kernel (gemm_f32.hip) pays a
pipeline fill (first operand chunk: a full L2 / HBM latency) and a drain per tile -- 91 TFLOP/s at 265 000 x 256 x 256 against
131 at K = 2 048.  Here a workgroup owns ONE column tile and STREAMS down its range of row tiles: the operand pipeline never
drains at a tile boundary (while the last chunks of tile t are multiplied, chunks 0 and 1 of tile t+1 are already on their way),
the accumulators are flushed (bias, optional relu, store) between two chunks, and the loop goes on.

Per 32-wide K chunk (64 x v_mfma_f32_32x32x2_f32 per wave, 4 waves = 2 x 2 of 64 x 64):
    wait the register stage (chunk g+1) | park it in the other LDS buffer | request chunk g+2 | read fragment set F1(g)
    32 MFMAs on F0(g), one memory instruction slotted behind each of the first 24, scalar pointer updates behind the rest
    s_waitcnt lgkmcnt(0); s_barrier
    read F0(g+1) from the other buffer behind the first 8 of the 32 MFMAs on F1(g)
LDS layout, fragment mapping and the k-ordering are those of gemm_nt_fast_kernel (plane / XOR layout, gemm_f32.hip).
All registers the statement touches are named literally and declared clobbered (see tools/gen_scan_mainloop.py for why).

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
ACC0 = 64                          # acc[i][j]: 16 registers each, v[64:127]
FRAG = {(0, 'A'): 128, (0, 'B'): 144, (1, 'A'): 160, (1, 'B'): 176}     # fragment set -> base; [q_ (2)][t (2)] float4
STAGE = 192                        # ra0..3, rb0..3: v[192:223]
V_LO, V_HI = 64, 223
# scalars
S_PA, S_PB, S_PC, S_ROW = 70, 72, 74, 88
S_TSTEP, S_KB, S_LDC4, S_LDC4X5, S_NT, S_NK2, S_LK, S_NKM1, S_T0, S_T1, S_LT, S_T2, S_CNT = 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 90
S_LO, S_HI = 70, 90


def v4(b):
    return "v[%d:%d]" % (b, b + 3)


def acc(i, j):
    b = ACC0 + (i * 2 + j) * 16
    return "v[%d:%d]" % (b, b + 15)


def lstore(buf):
    o = buf * BUF_BYTES
    ins = ["ds_write_b128 %%[ls0], %s offset:%d" % (v4(STAGE + 4 * s), o + s * 32 * 16) for s in range(4)]
    ins += ["ds_write_b128 %%[ls0], %s offset:%d" % (v4(STAGE + 16 + 4 * s), o + OPER_BYTES + s * 32 * 16) for s in range(4)]
    return ins


def gload():
    ins = []
    for s in range(4):     # A and B passes alternate like the C++ kernel does
        ins.append("global_load_dwordx4 %s, %%[oa%d], s[%d:%d]" % (v4(STAGE + 4 * s), s, S_PA, S_PA + 1))
        ins.append("global_load_dwordx4 %s, %%[ob%d], s[%d:%d]" % (v4(STAGE + 16 + 4 * s), s, S_PB, S_PB + 1))
    # advance the load stream by one chunk; at the end of a row tile step to the next one (or stay on the last one)
    adv = ["s_cmp_eq_u32 s%d, s%d" % (S_LK, S_NKM1),                       # SCC = this was the last chunk of its row tile
           "s_cselect_b32 s%d, s%d, 0" % (S_T1, S_KB),                     # rewind by K bytes ...
           "s_cselect_b32 s%d, s%d, 0" % (S_T2, S_TSTEP),                  # ... and step one row tile down
           "s_cselect_b32 s%d, -1, s%d" % (S_LK, S_LK),
           "s_add_u32 s%d, s%d, 1" % (S_LK, S_LK),
           "s_cmp_gt_u32 s%d, 1" % S_LT,                                   # is there another row tile to fetch?
           "s_cselect_b32 s%d, s%d, 0" % (S_T2, S_T2),
           "s_cmp_lg_u32 s%d, 0" % S_T2,
           "s_cselect_b32 s%d, 1, 0" % S_T0,
           "s_sub_u32 s%d, s%d, s%d" % (S_LT, S_LT, S_T0),
           "s_add_u32 s%d, s%d, 128" % (S_PA, S_PA), "s_addc_u32 s%d, s%d, 0" % (S_PA + 1, S_PA + 1),
           "s_sub_u32 s%d, s%d, s%d" % (S_PA, S_PA, S_T1), "s_subb_u32 s%d, s%d, 0" % (S_PA + 1, S_PA + 1),
           "s_add_u32 s%d, s%d, s%d" % (S_PA, S_PA, S_T2), "s_addc_u32 s%d, s%d, 0" % (S_PA + 1, S_PA + 1),
           "s_add_u32 s%d, s%d, 128" % (S_PB, S_PB), "s_addc_u32 s%d, s%d, 0" % (S_PB + 1, S_PB + 1),
           "s_sub_u32 s%d, s%d, s%d" % (S_PB, S_PB, S_T1), "s_subb_u32 s%d, s%d, 0" % (S_PB + 1, S_PB + 1)]
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


def mfmas(fset):
    """32 MFMAs of one fragment set, component-major: consecutive instructions never touch the same accumulator."""
    out = []
    for q_ in range(2):
        for c in range(4):
            for i in range(2):
                for j in range(2):
                    a = FRAG[(fset, 'A')] + (q_ * 2 + i) * 4 + c
                    b = FRAG[(fset, 'B')] + (q_ * 2 + j) * 4 + c
                    out.append("v_mfma_f32_32x32x2_f32 %s, v%d, v%d, %s" % (acc(i, j), a, b, acc(i, j)))
    return out


def slot(mf, mem):
    out = []
    for i, x in enumerate(mf):
        out.append(x)
        if i < len(mem):
            out.append(mem[i])
    return out + list(mem[len(mf):])


def head(nxt):
    """Start of a chunk: the stage (chunk g+1) has landed -> park it in the other LDS buffer, request chunk g+2."""
    ld, adv = gload()
    return ["s_waitcnt vmcnt(0) lgkmcnt(0)"], lstore(nxt) + ld, adv


def chunk(cur, nxt, with_head=True):
    """with_head=False: the first chunk of a row tile, whose head was issued in front of the previous tile's stores (epilogue)."""
    if with_head:
        ins, mem, adv = head(nxt)
    else:
        ins, mem, adv = ["s_waitcnt lgkmcnt(0)"], [], []
    ins += slot(mfmas(0), mem + fread(1, cur) + adv[:8])
    ins += adv[8:]
    ins += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    ins += slot(mfmas(1), fread(0, nxt))
    return ins


def epilogue(relu):
    """Flush the four 64 x 64 quadrants of this wave: + bias, [relu], store row by row, zero.  C/D layout of
    v_mfma_f32_32x32x2_f32: register r of lane (fi = lane & 31, fg = lane >> 5) is row (r & 3) + 8 (r >> 2) + 4 fg, column fi."""
    ins = ["s_nop 15", "s_nop 7"]                 # 16-pass MFMA results -> VALU
    for i in range(2):
        for j in range(2):
            for r in range(16):
                reg = ACC0 + (i * 2 + j) * 16 + r
                ins.append("v_add_f32 v%d, v%d, %%[bias%d]" % (reg, reg, j))
                if relu:
                    ins.append("v_max_f32 v%d, v%d, 0" % (reg, reg))
    # The head of the next chunk goes IN FRONT of the stores: vmcnt counts loads and stores in issue order and stops at 63, so a
    # wait for the operand loads issued before 64 stores would also wait for every store's write acknowledgement.
    w, mem, adv = head(1)
    ins += w + mem + adv
    ins += ["s_mov_b64 s[%d:%d], s[%d:%d]" % (S_ROW, S_ROW + 1, S_PC, S_PC + 1)]
    for idx in range(32):
        i, r = idx // 16, idx % 16
        for j in range(2):
            ins.append("global_store_dword %%[voffc], v%d, s[%d:%d] offset:%d" % (ACC0 + (i * 2 + j) * 16 + r, S_ROW, S_ROW + 1, j * 128))
        step = S_LDC4X5 if idx % 4 == 3 else S_LDC4
        ins += ["s_add_u32 s%d, s%d, s%d" % (S_ROW, S_ROW, step), "s_addc_u32 s%d, s%d, 0" % (S_ROW + 1, S_ROW + 1)]
    ins += ["v_mov_b32 v%d, 0" % r for r in range(ACC0, ACC0 + 64)]
    ins += ["s_lshl_b32 s%d, s%d, 7" % (S_T0, S_LDC4), "s_add_u32 s%d, s%d, s%d" % (S_PC, S_PC, S_T0),
            "s_addc_u32 s%d, s%d, 0" % (S_PC + 1, S_PC + 1)]
    return ins


def program(relu):
    L = lambda name: ".Lgs%d_%s_%%=" % (relu, name)
    ins = ["s_mov_b64 s[%d:%d], %%[pa]" % (S_PA, S_PA + 1), "s_mov_b64 s[%d:%d], %%[pb]" % (S_PB, S_PB + 1),
           "s_mov_b64 s[%d:%d], %%[pc]" % (S_PC, S_PC + 1), "s_mov_b32 s%d, %%[tstep]" % S_TSTEP, "s_mov_b32 s%d, %%[kbytes]" % S_KB,
           "s_mov_b32 s%d, %%[ldc4]" % S_LDC4, "s_mul_i32 s%d, s%d, 5" % (S_LDC4X5, S_LDC4), "s_mov_b32 s%d, %%[ntile]" % S_NT,
           "s_mov_b32 s%d, %%[ntile]" % S_LT, "s_mov_b32 s%d, %%[nk2]" % S_NK2, "s_lshl_b32 s%d, s%d, 1" % (S_NKM1, S_NK2),
           "s_sub_u32 s%d, s%d, 1" % (S_NKM1, S_NKM1), "s_mov_b32 s%d, 0" % S_LK]
    ins += ["v_mov_b32 v%d, 0" % r for r in range(ACC0, ACC0 + 64)]
    # prologue: chunk 0 -> LDS buffer 0, chunk 1 -> stage
    ld, adv = gload()
    ins += ld + adv + ["s_waitcnt vmcnt(0)"] + lstore(0)
    ld, adv = gload()
    ins += ld + adv + ["s_waitcnt lgkmcnt(0)", "s_barrier"] + fread(0, 0)
    w, mem, adv = head(1)                          # head of the first tile's first chunk (later tiles: inside the epilogue)
    ins += w + mem + adv
    ins += [L("tile") + ":"]
    ins += chunk(0, 1, with_head=False) + chunk(1, 0)
    ins += ["s_sub_u32 s%d, s%d, 1" % (S_CNT, S_NK2), "s_cmp_eq_u32 s%d, 0" % S_CNT, "s_cbranch_scc1 " + L("flush"), L("chunk") + ":"]
    ins += chunk(0, 1) + chunk(1, 0)
    ins += ["s_sub_u32 s%d, s%d, 1" % (S_CNT, S_CNT), "s_cmp_lg_u32 s%d, 0" % S_CNT, "s_cbranch_scc1 " + L("chunk")]
    ins += [L("flush") + ":"]
    ins += epilogue(relu)
    ins += ["s_sub_u32 s%d, s%d, 1" % (S_NT, S_NT), "s_cmp_lg_u32 s%d, 0" % S_NT, "s_cbranch_scc1 " + L("tile")]
    ins += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]      # the tail prefetch (re-read of the last tile) and the stores are done
    return ins


def render_one(relu):
    ins = program(relu)
    lines = ["    asm volatile("]
    for x in ins:
        lines.append('        "%s\\n\\t"' % x)
    lines.append("        :")
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
           "// Body of gemm_nt_stream_kernel<RELU>: prologue + tile loop (chunk loop, flush) with hand-allocated registers.",
           "    if constexpr (RELU) {", render_one(1).rstrip("\n"), "    } else {", render_one(0).rstrip("\n"), "    }"]
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    open(OUT, "w").write(text)
    print("wrote %s: %d lines" % (OUT, text.count("\n")))
