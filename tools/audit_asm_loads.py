#!/usr/bin/env python3
"""Build-time audit of the hand-counted inline-asm loads (csrc/scan_mainloop*.inc, gemm_f32.hip, sgraf_loc.hip).

Those kernels issue `global_load_dwordx4` through `asm volatile` so that hipcc's s_waitcnt bookkeeping does not see
them, and wait for them with their own `s_waitcnt vmcnt(N)` statements.  hipcc treats the destination of such a load
as written when the asm statement ends, so it is free to copy, spill or REUSE that register while the data is still in
flight (cdna_hip_programming.md 5.7 item 1) -- silent corruption that depends on register allocation, i.e. on the
compiler version and on unrelated edits.  This script compiles a source with -save-temps and proves, on the generated
ISA, for every kernel that contains asm loads:

  1. between an asm load and the asm wait that covers its destination, no compiler-generated instruction reads or writes
     that register (forward dataflow over the kernel's control-flow graph, union at joins);
  2. nothing is in flight at s_endpgm;
  3. (optional, --no-scratch-in-loops) no scratch (spill) instruction sits in a basic block that is part of a loop which also
     contains asm loads: spill traffic shares the vmcnt counter with the hand-counted loads.  A kernel whose asm waits are ALL
     `vmcnt(0)` (sgr_fused.hip) counts nothing by hand -- extra memory operations in flight cannot make a `vmcnt(0)` return early --
     so there the rule is the weaker, sufficient one: no scratch instruction while a global asm load is in flight.

A wait statement covers the registers it prints in its trailing comment (`s_waitcnt vmcnt(7) ; covers v[1:4] ...`, the
"+v" operands of the statement); a bare `s_waitcnt vmcnt(0)` inside an asm block covers everything.

    python tools/audit_asm_loads.py image-text-retrieval_amd/csrc/scan_xattn.hip [--kernel SUBSTR] [--no-scratch-in-loops]
Exit code 0 = clean.  Used by tests/test_isa_audit.py (CPU: hipcc cross-compiles without a GPU).
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.environ.get("ITR_CSRC", os.path.join(ROOT, "image-text-retrieval_amd", "csrc"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

_REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def compile_to_asm(src, workdir):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-save-temps", "-c", src, "-I", CSRC,
           "-I", os.path.join(ROOT, "include"), "-o", os.path.join(workdir, "out.o")]
    subprocess.run(cmd, cwd=workdir, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    for f in os.listdir(workdir):
        if f.endswith("gfx950.s"):
            return os.path.join(workdir, f)
    raise RuntimeError("no device assembly produced")


def split_kernels(asm_path):
    """-> {kernel symbol: [lines]} for every .amdhsa kernel in the file."""
    kernels, cur, name = {}, None, None
    for line in open(asm_path):
        m = re.match(r"^(_Z\w+):\s", line)
        if m and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line.rstrip("\n"))
            if ".end_amdhsa_kernel" in line or line.startswith(".Lfunc_end"):
                kernels[name] = cur
                cur, name = None, None
    return kernels


class Block:
    def __init__(self, label):
        self.label, self.insts, self.succ = label, [], []


def build_cfg(lines):
    """Basic blocks of one kernel.  insts: (kind, text) with kind in {'asm', 'cc'} (inside / outside ASMSTART..ASMEND)."""
    blocks, cur, in_asm = [], Block("entry"), False
    blocks.append(cur)
    pending_fall = True
    for raw in lines:
        s = raw.strip()
        if not s:
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            nb = Block(m.group(1))
            if pending_fall:
                cur.succ.append(nb.label)
            blocks.append(nb)
            cur, pending_fall = nb, True
            continue
        if s.startswith(";") or s.startswith("."):
            continue
        cur.insts.append(("asm" if in_asm else "cc", s))
        op = s.split()[0]
        if op == "s_branch":
            cur.succ.append(s.split()[1])
            nb = Block("after_%d" % len(blocks))
            blocks.append(nb)
            cur, pending_fall = nb, False      # unreachable by fall-through; only via a label (it will be relabelled)
            pending_fall = False
        elif op.startswith("s_cbranch"):
            cur.succ.append(s.split()[1])
            nb = Block("ft_%d" % len(blocks))
            cur.succ.append(nb.label)
            blocks.append(nb)
            cur, pending_fall = nb, True
        elif op == "s_endpgm":
            nb = Block("dead_%d" % len(blocks))
            blocks.append(nb)
            cur, pending_fall = nb, False
        elif op in ("s_setpc_b64", "s_swappc_b64"):
            pass
    return blocks


class Block1:
    """a one-instruction view of a block (for walking a block instruction by instruction with transfer())"""
    def __init__(self, label, insts):
        self.label, self.insts = label, insts


def transfer(block, inflight, report):
    """Walk one block.  inflight: {reg: 'where the load was issued'}.  Returns the state at the block's end."""
    st = dict(inflight)
    for kind, s in block.insts:
        code = s.split(";")[0]
        op = code.split()[0]
        if kind == "asm":
            if op.startswith("global_load") or op.startswith("buffer_load"):
                if " lds" in code or "_lds_" in op:
                    continue                      # LDS-DMA: no VGPR destination
                dst = regs_of(code.split(",")[0])
                for r in dst:
                    st[r] = "%s: %s" % (block.label, code.strip())
            elif op.startswith("ds_read"):        # asm LDS reads (sgr_fused.hip): in flight until an asm lgkmcnt(0)
                for r in regs_of(code.split(",")[0]):
                    st[r] = "lds %s: %s" % (block.label, code.strip())
            elif op == "s_waitcnt" and ("vmcnt" in code or "lgkmcnt" in code):
                cov = s.split("covers", 1)[1] if "covers" in s else None
                if cov is None:
                    if "vmcnt(0)" in code:
                        for r in [r for r, w in st.items() if not w.startswith("lds ")]:
                            st.pop(r)
                    if "lgkmcnt(0)" in code:
                        for r in [r for r, w in st.items() if w.startswith("lds ")]:
                            st.pop(r)
                else:
                    for r in regs_of(cov):
                        st.pop(r, None)
            continue
        if op == "s_endpgm":
            if st:
                report.append("in flight at s_endpgm: v%s" % sorted(st))
            continue
        touched = regs_of(code) & set(st)
        if touched:
            report.append("%s: `%s` touches v%s while its asm load is in flight (%s)" % (
                block.label, code.strip(), sorted(touched), st[sorted(touched)[0]]))
    return st


def audit_kernel(name, lines, no_scratch_in_loops=False):
    blocks = build_cfg(lines)
    by_label = {b.label: b for b in blocks}
    if not any(k == "asm" and (s.startswith("global_load") or s.startswith("buffer_load")) for b in blocks for k, s in b.insts):
        return None
    state_in = {b.label: {} for b in blocks}
    work = [blocks[0].label]
    seen_out = {}
    n_iter = 0
    while work:
        n_iter += 1
        if n_iter > 200000:
            raise RuntimeError("dataflow does not converge")
        lab = work.pop()
        b = by_label[lab]
        out = transfer(b, state_in[lab], [])
        if seen_out.get(lab) == out:
            continue
        seen_out[lab] = out
        for s in b.succ:
            if s not in by_label:
                continue
            merged = dict(state_in[s])
            merged.update(out)
            if merged != state_in[s] or s not in seen_out:
                state_in[s] = merged
                work.append(s)
    report = []
    for b in blocks:
        transfer(b, state_in[b.label], report)
    asm_waits = [s for b in blocks for k, s in b.insts if k == "asm" and s.split()[0] == "s_waitcnt" and "vmcnt" in s.split(";")[0]]
    only_full_waits = bool(asm_waits) and all("vmcnt(0)" in s.split(";")[0] for s in asm_waits)
    if no_scratch_in_loops and only_full_waits:
        for b in blocks:
            st = dict(state_in[b.label])
            for kind, s in b.insts:
                if kind == "cc" and s.startswith("scratch_") and any(not w.startswith("lds ") for w in st.values()):
                    report.append("%s: spill `%s` while an asm load is in flight" % (b.label, s.split(";")[0].strip()))
                st = transfer(Block1(b.label, [(kind, s)]), st, [])
    elif no_scratch_in_loops:
        # blocks on a cycle that also holds asm loads
        idx = {b.label: i for i, b in enumerate(blocks)}
        reach = {}

        def reachable(src):
            if src in reach:
                return reach[src]
            seen, stack = set(), [src]
            while stack:
                x = stack.pop()
                for s in by_label[x].succ:
                    if s in by_label and s not in seen:
                        seen.add(s)
                        stack.append(s)
            reach[src] = seen
            return seen
        for b in blocks:
            if b.label in reachable(b.label):          # on a cycle
                cyc = [c for c in reachable(b.label) if b.label in reachable(c)]
                has_asm_load = any(k == "asm" and s.startswith(("global_load", "buffer_load")) for c in cyc for k, s in by_label[c].insts)
                if has_asm_load:
                    for k, s in b.insts:
                        if s.startswith("scratch_"):
                            report.append("%s: spill `%s` inside a loop with hand-counted asm loads" % (b.label, s.split(";")[0].strip()))
    # de-duplicate, keep order
    out, seen = [], set()
    for r in report:
        if r not in seen:
            seen.add(r)
            out.append(r)
    return out


def audit_file(src, kernel_filter=None, no_scratch_in_loops=False, keep=None):
    with tempfile.TemporaryDirectory() as tmp:
        asm = compile_to_asm(os.path.abspath(src), tmp)
        if keep:
            import shutil
            shutil.copy(asm, keep)
        results = {}
        for name, lines in split_kernels(asm).items():
            if kernel_filter and kernel_filter not in name:
                continue
            r = audit_kernel(name, lines, no_scratch_in_loops)
            if r is not None:
                results[name] = r
        return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--kernel", default=None, help="only kernels whose mangled name contains this")
    ap.add_argument("--no-scratch-in-loops", action="store_true")
    ap.add_argument("--keep-asm", default=None)
    a = ap.parse_args()
    res = audit_file(a.source, a.kernel, a.no_scratch_in_loops, a.keep_asm)
    bad = 0
    for name, rep in res.items():
        print("%s: %s" % (name, "clean" if not rep else "%d finding(s)" % len(rep)))
        for r in rep[:40]:
            print("    " + r)
        bad += len(rep)
    if not res:
        print("no kernel with asm loads found")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
