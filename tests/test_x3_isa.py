"""CPU (hipcc cross-compiles gfx950 without a GPU): the counted stage waits of the 32-point-wave bf16x3 kernels, checked on the ISA.

The K-outer ops (i2sdf_amd/csrc/x3.h: dense_x3g) hand every weight stage over with `s_waitcnt vmcnt(N); s_barrier`, N = the number of
vector-memory instructions the wave has issued UNCONDITIONALLY since the stage's last DMA piece (common.h: WStreamT::advance_barrier_young).
N is an ordinary `int` that only becomes an immediate because every loop around it is fully unrolled; if an edit or a compiler update leaves
it a run-time value, the `switch` in advance_barrier_young stays a jump table (slow) -- and if a source's load / store count and the
instructions actually emitted ever disagree in the unsafe direction, a stage could be read before its DMA has landed.  Checked here:
  1. no scratch in the three kernels (a spill's reload is a vector-memory instruction nobody counted: harmless for safety -- under-counting
     only waits longer -- but it drains the pipeline),
  2. the counted barriers are single asm statements `s_waitcnt vmcnt(N); s_barrier` with an immediate N (no jump table),
  3. a good share of them have N > 0 (the optimisation is alive), and
  4. N never exceeds the vector-memory instructions emitted between the previous stage's barrier and this one that can be young at all."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "i2sdf_amd", "csrc")
KERNELS = {"sdf_igrad3_kernel": 6, "sdf_bwd3_sweep1_kernel": 12, "sdf_bwd3_sweep2_kernel": 12}      # name -> minimum number of counted barriers


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    out = tmp_path_factory.mktemp("isa") / "mlp_x3.s"
    # the flags of i2sdf_amd/csrc/build.sh for mlp_x3.hip
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-Wno-unused-result",
           "-mllvm", "-pragma-unroll-threshold=1000000", "-x", "hip", "--cuda-device-only", "-S", os.path.join(CSRC, "mlp_x3.hip"), "-o", str(out)]
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


def _bodies(text):
    """{mangled name: [lines]} of every kernel in the file"""
    out, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1); out[cur] = []
        elif cur is not None:
            if ln.startswith(".Lfunc_end"):
                cur = None
            else:
                out[cur].append(ln)
    return out


def _insn(ln):
    ln = ln.split(";")[0].strip()
    return ln.split()[0] if ln else ""


VMEM = re.compile(r"^(global_load|global_store|buffer_load|buffer_store|scratch_load|scratch_store|flat_load|flat_store)")


def test_stage_barriers_carry_counted_waits(isa):
    bodies = _bodies(isa)
    for key, min_counted in KERNELS.items():
        names = [n for n in bodies if key in n]
        assert names, f"{key} not found in the ISA"
        for name in names:
            body = bodies[name]
            counted, since_barrier, nbar, nasm, in_asm = 0, 0, 0, 0, False
            last_wait = None
            for ln in body:
                if "#ASMSTART" in ln:
                    in_asm, last_wait = True, None
                    continue
                if "#ASMEND" in ln:
                    in_asm = False
                    continue
                op = _insn(ln)
                if not op:
                    continue
                if VMEM.match(op):
                    since_barrier += 1
                if in_asm:
                    m = re.match(r"\s*s_waitcnt\s+vmcnt\((\d+)\)", ln)
                    if m:
                        last_wait = int(m.group(1))
                if op == "s_barrier":
                    nbar += 1
                    if in_asm:      # the counted form (common.h: wait_barrier<N>): one asm statement, the wait directly in front of the barrier
                        nasm += 1
                        assert last_wait is not None, f"{name}: barrier #{nbar}: asm statement without its s_waitcnt"
                        # whatever may still be in flight was issued after the previous barrier (everything older was drained or counted there)
                        assert last_wait <= since_barrier, \
                            f"{name}: barrier #{nbar} allows {last_wait} instructions in flight, only {since_barrier} were issued since the last barrier"
                        counted += last_wait > 0
                    since_barrier = 0
            assert nbar >= 20 and nasm >= nbar // 2, (name, nbar, nasm)
            assert counted >= min_counted, f"{name}: only {counted} of {nbar} stage barriers carry a counted wait"
            # no jump table left from advance_barrier_young's switch
            assert not any("s_setpc_b64" in ln for ln in body), f"{name}: indirect branch in the kernel body"


def test_no_scratch_in_the_counted_kernels(isa):
    for key in KERNELS:
        for m in re.finditer(r"\.name:\s+(\S*%s\S*)\n(?:.*\n){0,12}?\s+\.private_segment_fixed_size:\s+(\d+)" % key, isa):
            assert int(m.group(2)) == 0, (m.group(1), m.group(2))
        sizes = re.findall(r"\.private_segment_fixed_size:\s+(\d+)", isa)
        assert sizes, "no kernel metadata found"
