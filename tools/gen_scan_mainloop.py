#!/usr/bin/env python3
"""Generates image-text-retrieval_amd/csrc/scan_mainloop_asm.inc: the fp32 main loop of the SCAN kernel as ONE inline-asm
statement with hand-allocated registers.

Why assembly.  The loop issues its global loads two chunks ahead and counts them with its own s_waitcnt vmcnt(N) (hipcc's
bookkeeping exposed a full memory latency per chunk).  A load hipcc cannot see has, to hipcc, a destination that is
"written" when the asm statement ends: with compiler-allocated stage registers the allocator is free to copy, spill or reuse
them while data is in flight, and at this kernel's register budget every restructuring of the loop made it do so
(tools/audit_asm_loads.py).  Here every register the loop touches is named literally and listed as clobbered, so hipcc
keeps its own values out of them for the whole statement, nothing is in flight when the statement ends (the last wait is
vmcnt(0)), and the instruction interleave is exactly the one written below instead of a scheduler's best effort.

Loop structure (per 32-wide K chunk, 72 x v_mfma_f32_16x16x4_f32 per wave):
    wait stage P (chunk kc+1) | park it in the other LDS buffer | refill P with chunk kc+3 | read fragment set F1(kc)
    36 MFMAs on F0(kc), one memory instruction slotted behind each of the first 24
    s_waitcnt lgkmcnt(0); s_barrier            (chunk kc+1 visible; every wave done reading chunk kc's F1 half)
    read F0(kc+1) from the other buffer, slotted behind the first 10 of the 36 MFMAs on F1(kc)
The last three chunks take no refill; the very last parks nothing.  Then the accumulators are parked transposed in LDS.

    python tools/gen_scan_mainloop.py            # rewrite the .inc
    python tools/gen_scan_mainloop.py --check    # exit 1 if the committed .inc differs from what this script generates
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "image-text-retrieval_amd", "csrc", "scan_mainloop_asm.inc")

# ---- geometry (scan_common.h)
SC_MT, SC_NT, SC_MTILES = 144, 64, 9
SC_ROWS, SC_PLANES = SC_MT + SC_NT + 16, 8
STAGE_BYTES = SC_PLANES * SC_ROWS * 16          # 28 672: one LDS operand buffer
SC_LDT = SC_MT + 4                              # row stride (floats) of the parked block [column][row]

# ---- register map (VGPRs 64..235 and SGPRs 80..86 belong to the statement)
ACC0, F0A, F0B, F1A, F1B, STA, STB = 64, 100, 136, 140, 176, 180, 208
V_LO, V_HI = 64, 235
S_PA, S_PB, S_PB2, S_N = 80, 82, 84, 86


def vr(base, n=4):
    return "v[%d:%d]" % (base, base + n - 1)


def acc(m):
    return vr(ACC0 + 4 * m)


def stage(P, i):
    return vr((STA if P == 'A' else STB) + 4 * i)


def lstore(P, buf):
    """7 x ds_write_b128: stage P -> LDS buffer `buf` (plane / XOR layout; pass c of rows = la0 + 16 c, see scan_mainloop.inc)."""
    o = buf * STAGE_BYTES
    return ["ds_write_b128 %%[la0], %s offset:%d" % (stage(P, 0), o),
            "ds_write_b128 %%[la0], %s offset:%d" % (stage(P, 1), o + 32 * 16),
            "ds_write_b128 %%[la0], %s offset:%d" % (stage(P, 2), o + 64 * 16),
            "ds_write_b128 %%[la0], %s offset:%d" % (stage(P, 3), o + 96 * 16),
            "ds_write_b128 %%[la4], %s offset:%d" % (stage(P, 4), o),
            "ds_write_b128 %%[la0], %s offset:%d" % (stage(P, 5), o + SC_MT * 16),
            "ds_write_b128 %%[la0], %s offset:%d" % (stage(P, 6), o + (SC_MT + 32) * 16)]


def gload(P):
    """7 x global_load_dwordx4 of the chunk the running pointers address, then advance the pointers by one chunk (128 B)."""
    ins = ["global_load_dwordx4 %s, %%[va%d], s[%d:%d]" % (stage(P, i), i, S_PA, S_PA + 1) for i in range(5)]
    ins.append("global_load_dwordx4 %s, %%[vb0], s[%d:%d]" % (stage(P, 5), S_PB, S_PB + 1))
    ins.append("global_load_dwordx4 %s, %%[vb0], s[%d:%d]" % (stage(P, 6), S_PB2, S_PB2 + 1))
    adv = []
    for s in (S_PA, S_PB, S_PB2):
        adv += ["s_add_u32 s%d, s%d, 128" % (s, s), "s_addc_u32 s%d, s%d, 0" % (s + 1, s + 1)]
    return ins, adv


def fread(which, buf):
    """10 x ds_read_b128: fragment set F0 (k-planes 0..3) or F1 (planes 4..7) of the chunk in LDS buffer `buf`."""
    fa, fb, ra, rb = (F0A, F0B, "ra0", "rb0") if which == 0 else (F1A, F1B, "ra1", "rb1")
    o = buf * STAGE_BYTES
    return ["ds_read_b128 %s, %%[%s] offset:%d" % (vr(fb), rb, o)] + \
           ["ds_read_b128 %s, %%[%s] offset:%d" % (vr(fa + 4 * m), ra, o + m * 256) for m in range(SC_MTILES)]


def mfmas(which):
    """36 MFMAs of one fragment set, component-major (consecutive MFMAs hit different accumulators)."""
    fa, fb = (F0A, F0B) if which == 0 else (F1A, F1B)
    return ["v_mfma_f32_16x16x4_f32 %s, v%d, v%d, %s" % (acc(m), fa + 4 * m + c, fb + c, acc(m))
            for c in range(4) for m in range(SC_MTILES)]


def chunk(P, cur, nxt, vmn, load):
    """One steady-state / tail chunk.  P: the stage that holds chunk kc+1; cur / nxt: LDS buffers of chunk kc / kc+1."""
    ins = ["s_waitcnt vmcnt(%d) lgkmcnt(0)" % vmn]        # stage P landed; the F0 fragments read during the previous half landed
    mem = lstore(P, nxt)
    adv = []
    if load:
        ld, adv = gload(P)
        mem += ld
    mem += fread(1, cur)
    half1 = []
    mf = mfmas(0)
    for i, x in enumerate(mf):
        half1.append(x)
        if i < len(mem):
            half1.append(mem[i])
        elif adv:
            half1.append(adv.pop(0))
    ins += half1
    ins += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    mem2 = fread(0, nxt)
    for i, x in enumerate(mfmas(1)):
        ins.append(x)
        if i < len(mem2):
            ins.append(mem2[i])
    return ins


def last(cur):
    ins = ["s_waitcnt lgkmcnt(0)"]
    mem = fread(1, cur)
    for i, x in enumerate(mfmas(0)):
        ins.append(x)
        if i < len(mem):
            ins.append(mem[i])
    ins.append("s_waitcnt lgkmcnt(0)")
    ins += mfmas(1)
    return ins


def program():
    L = lambda name: ".Lscan_%s_%%=" % name          # %= : unique per asm statement instance
    ins = []
    ins += ["s_mov_b64 s[%d:%d], %%[abase]" % (S_PA, S_PA + 1), "s_mov_b64 s[%d:%d], %%[bbase]" % (S_PB, S_PB + 1),
            "s_mov_b64 s[%d:%d], %%[bbase2]" % (S_PB2, S_PB2 + 1), "s_mov_b32 s%d, %%[nmain]" % S_N]
    ins += ["v_mov_b32 v%d, 0" % r for r in range(ACC0, ACC0 + 4 * SC_MTILES)]
    # prologue: chunk 0 -> LDS buffer 0, stage B = chunk 1, stage A = chunk 2
    for P in ('A', 'B'):
        ld, adv = gload(P)
        ins += ld + adv
    ins += ["s_waitcnt vmcnt(7)"] + lstore('A', 0)
    ld, adv = gload('A')
    ins += ld + adv
    ins += ["s_waitcnt lgkmcnt(0)", "s_barrier"] + fread(0, 0)
    # steady state: chunks 0 .. nk-4; s86 counts the chunks left that still have a chunk kc+3 to fetch
    ins += [L("loop") + ":",
            "s_cmp_lt_i32 s%d, 1" % S_N, "s_cbranch_scc1 " + L("tail_even")]
    ins += chunk('B', 0, 1, 7, True)
    ins += ["s_sub_u32 s%d, s%d, 1" % (S_N, S_N), "s_cmp_lt_i32 s%d, 1" % S_N, "s_cbranch_scc1 " + L("tail_odd")]
    ins += chunk('A', 1, 0, 7, True)
    ins += ["s_sub_u32 s%d, s%d, 1" % (S_N, S_N), "s_branch " + L("loop")]
    # peeled tails: chunks nk-3, nk-2 (no refill; the second wait drains everything), nk-1
    ins += [L("tail_even") + ":"] + chunk('B', 0, 1, 7, False) + chunk('A', 1, 0, 0, False) + last(0) + ["s_branch " + L("done")]
    ins += [L("tail_odd") + ":"] + chunk('A', 1, 0, 7, False) + chunk('B', 0, 1, 0, False) + last(1)
    ins += [L("done") + ":"]
    # park the 144 x 64 block transposed: arawt[column][row]; the staging buffers alias it, so every wave must be done reading
    ins += ["s_nop 15", "s_nop 3", "s_barrier"]               # MFMA results readable (8-pass XDL -> LDS data read), all F reads done
    ins += ["ds_write_b128 %%[park], %s offset:%d" % (acc(m), m * 64) for m in range(SC_MTILES)]
    ins += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    return ins


def render():
    ins = program()
    lines = []
    lines.append("// GENERATED by tools/gen_scan_mainloop.py -- do not edit; regenerate and commit (tests/test_isa_audit.py checks it is current).")
    lines.append("// One asm statement = prologue + K loop + peeled tail + park of scan_xattn_body<0>; register map and rationale in the generator.")
    lines.append("    asm volatile(")
    for x in ins:
        lines.append('        "%s\\n\\t"' % x)
    lines.append("        :")
    lines.append('        : [va0] "v"(va0), [va1] "v"(va1), [va2] "v"(va2), [va3] "v"(va3), [va4] "v"(va4), [vb0] "v"(vb0), [la0] "v"(la0), [la4] "v"(la4),')
    lines.append('          [ra0] "v"(ra0), [ra1] "v"(ra1), [rb0] "v"(rb0), [rb1] "v"(rb1), [park] "v"(park_addr),')
    lines.append('          [abase] "s"(abase), [bbase] "s"(bbase), [bbase2] "s"(bbase2), [nmain] "s"(nmain)')
    clob = ['"memory"', '"scc"'] + ['"s%d"' % s for s in range(S_PA, S_N + 1)] + ['"v%d"' % r for r in range(V_LO, V_HI + 1)]
    row, rows = [], []
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
    n_mfma = text.count("v_mfma")
    print("wrote %s: %d lines, %d MFMAs (= 8 chunk bodies x 72 - the last-chunk halves)" % (OUT, text.count("\n"), n_mfma))
