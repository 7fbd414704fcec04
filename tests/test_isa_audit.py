"""CPU build check: the kernels that hide inline-asm loads from hipcc (hand-counted s_waitcnt vmcnt) are audited on the
ISA hipcc generates HERE (tools/audit_asm_loads.py): no compiler instruction may touch the destination of an asm load
before the wait that covers it, no spill may land in a loop with hand-counted loads, nothing may be in flight at
s_endpgm.  hipcc cross-compiles gfx950 without a GPU; ~15 s per source file."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import audit_asm_loads as A  # noqa: E402

CSRC = os.path.join(ROOT, "image-text-retrieval_amd", "csrc")
# (source, kernels that must be present and clean).  Not listed: scan_xattn_kernel<5> / <9> (ablation builds of tools/) and <3>
# (opt-in fp16x3 study variant: it drains inside its LAST chunk under `if (kc >= klast)`, which is correlated with the loop
# exit -- a union-at-joins dataflow cannot prove that, it reports the park code after the loop).
CASES = [("scan_xattn.hip", ["scan_xattn_kernelILi0E", "scan_xattn_kernelILi1E"]),
         ("gemm_f32.hip", ["gemm_nt_fast_kernel"]),
         ("gemm_stream.hip", ["gemm_nt_stream_kernelILi0E", "gemm_nt_stream_kernelILi1E", "gemm_nt_stream_kernelILi4E"]),
         ("sgraf_loc.hip", ["sgraf_loc_kernel"]),
         ("sgr_fused.hip", ["sgr_fused_kernel", "sgr_fused_persistent_kernel"])]       # weight fragments AND LDS node fragments through asm (SF_GLOAD / SF_LREAD)


@pytest.mark.skipif(shutil.which(A.HIPCC) is None and not os.path.exists(A.HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src,kernels", CASES)
def test_asm_loads_are_never_touched_in_flight(src, kernels):
    res = A.audit_file(os.path.join(CSRC, src), no_scratch_in_loops=True)
    for k in kernels:
        hits = [name for name in res if k in name]
        assert hits, "kernel %s not found in %s (or it no longer contains asm loads)" % (k, src)
        for name in hits:
            assert res[name] == [], "%s:\n  %s" % (name, "\n  ".join(res[name][:20]))


def test_audit_detects_a_touched_register():
    """The checker itself: a compiler v_mov of an in-flight destination and a load left in flight at s_endpgm are reported,
    a covered register is not."""
    lines = """
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v1, s[2:3]
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[14:17], v1, s[4:5]
	;;#ASMEND
	v_add_u32_e32 v20, v21, v22
	s_cbranch_scc1 .LBB0_2
	v_mov_b32_e32 v30, v12
.LBB0_2:
	;;#ASMSTART
	s_waitcnt vmcnt(1) ; covers v[10:13]
	;;#ASMEND
	v_mov_b32_e32 v31, v10
	v_mov_b32_e32 v32, v15
	s_endpgm
""".split("\n")
    rep = A.audit_kernel("k", lines)
    assert any("v30, v12" in r for r in rep)            # copy before the wait, on one path only
    assert any("v32, v15" in r for r in rep)            # second load never covered
    assert not any("v31, v10" in r for r in rep)        # covered by the wait
    assert any("in flight at s_endpgm" in r for r in rep)


@pytest.mark.parametrize("gen", ["gen_scan_mainloop.py", "gen_gemm_stream.py", "gen_sgraf_loc.py", "gen_gemm_tile.py"])
def test_generated_asm_bodies_are_current(gen):
    """csrc/scan_mainloop_asm.inc, gemm_stream_asm.inc and sgraf_loc_asm.inc are generated; the committed files must be what the generators emit."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen), "--check"])
    assert r.returncode == 0, "regenerate: python tools/%s" % gen
