"""CPU: the counted stage waits of the 32-point-wave bf16x3 kernels, checked on the ISA (disassembly of the in-tree build: seconds; without
one, or with a stale one, hipcc compiles mlp_x3.hip -- it cross-compiles gfx950 without a GPU -- which takes two minutes).

The K-outer ops (i2sdf_amd/csrc/x3.h: dense_x3g) hand every weight stage over with `s_waitcnt vmcnt(N); s_barrier`, N = the number of
vector-memory instructions the wave has issued UNCONDITIONALLY since the stage's last DMA piece (common.h: WStreamT::advance_barrier_young).
N is an ordinary `int` that only becomes an immediate because every loop around it is fully unrolled; if an edit or a compiler update leaves
it a run-time value, the `switch` in advance_barrier_young stays a jump table (slow) -- and if a source's load / store count and the
instructions actually emitted ever disagree in the unsafe direction, a stage could be read before its DMA has landed.  Checked here:
  1. no scratch in the three kernels (a spill's reload is a vector-memory instruction nobody counted: harmless for safety -- under-counting
     only waits longer -- but it drains the pipeline),
  2. the counted barriers are single asm statements `s_waitcnt vmcnt(N); s_barrier` with an immediate N (no jump table),
  3. a good share of them have N > 0 (the optimisation is alive), and
  4. N never exceeds the vector-memory instructions EMITTED BEHIND THE STAGE'S LAST DMA PIECE (`buffer_load ... lds`) -- the invariant the
     protocol needs (round 6; the round-5 form of this check counted from the previous barrier, which includes the pieces and whatever was
     issued before them: a plain global access hoisted above the last piece by the scheduler would have gone unnoticed -- ADVICE r5; the
     sources now fence the last piece with sched_barrier(0)), in the 32-point kernels (mlp_x3.o) AND the 16-point ones (mlp_x3h.o)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "i2sdf_amd", "csrc")
KERNELS = {"sdf_igrad3_kernel": 6, "sdf_bwd3_sweep1_kernel": 12, "sdf_bwd3_sweep2_kernel": 12}      # name -> minimum number of barriers with N > 0
# the 16-point-wave kernels (x3h.h: dense_x3h): the sources that issue global loads ahead (radiance backward: saved activations; radiance forward:
# feature rows; light head) carry N = 2 waits, the forward kernels' sources issue nothing unconditionally (N = 0 everywhere)
KERNELS_H = {"rgb_bwd3h_kernel": 6, "rgb_fwd3h_kernel": 1, "light_fwd3h_kernel": 1, "sdf_train_fwd3h_kernel": 0, "sdf_fwd3h_kernel": 0}


LLVM = "/opt/rocm/lib/llvm/bin"
OBJ = os.path.join(ROOT, "i2sdf_amd", "lib", "obj", "mlp_x3.o")
OBJ_H = os.path.join(ROOT, "i2sdf_amd", "lib", "obj", "mlp_x3h.o")


def _disassemble_in_tree(tmp, OBJ=OBJ, src="mlp_x3.hip"):
    """The device code of the in-tree build (seconds); None if there is none or it is older than the sources."""
    if not os.path.exists(OBJ) or not all(os.path.exists(f"{LLVM}/{t}") for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")):
        return None
    import glob
    if any(os.path.getmtime(f) > os.path.getmtime(OBJ) for f in glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(CSRC, src)]):
        return None
    fat, co = os.path.join(tmp, "x3.fatbin"), os.path.join(tmp, "x3.co")
    if subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", OBJ], capture_output=True).returncode != 0:
        return None
    if subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}",
                       "--unbundle"], capture_output=True).returncode != 0:
        return None
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True)
    if dis.returncode != 0:
        return None
    # objdump form -> the assembler-listing form the checks below read: "<name>:" labels, one instruction per line, no addresses
    lines = []
    for ln in dis.stdout.splitlines():
        m = re.match(r"^[0-9a-f]+ <(_Z\w+)>:", ln)
        if m:
            lines.append(".Lfunc_end:"); lines.append(m.group(1) + ":")
        elif ln.startswith("\t"):
            lines.append(ln.split("//")[0].rstrip())
    return "\n".join(lines) + "\n.Lfunc_end:\n" + notes.stdout


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    """(text, from_objdump).  The in-tree object when it is there and fresh; else a compile of mlp_x3.hip (~2 minutes)."""
    tmp = tmp_path_factory.mktemp("isa")
    text = _disassemble_in_tree(str(tmp))
    if text is not None:
        return text, True
    if shutil.which("hipcc") is None:
        pytest.skip("no in-tree build and no hipcc")
    out = tmp / "mlp_x3.s"
    # the flags of i2sdf_amd/csrc/build.sh for mlp_x3.hip
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-Wno-unused-result",
           "-mllvm", "-pragma-unroll-threshold=1000000", "-x", "hip", "--cuda-device-only", "-S", os.path.join(CSRC, "mlp_x3.hip"), "-o", str(out)]
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read(), False


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
WAIT = re.compile(r"\s*s_waitcnt\s+vmcnt\((\d+)\)(\s+lgkmcnt\(0\))?\s*$")


def _check_counted_barriers(bodies, kernels, from_objdump, min_barriers=20):
    for key, min_counted in kernels.items():
        names = [n for n in bodies if key in n]
        assert names, f"{key} not found in the ISA"
        for name in names:
            body = bodies[name]
            counted, since_piece, nbar, nasm, in_asm, npieces = 0, 0, 0, 0, False, 0
            last_wait, prev_wait = None, None
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
                code = ln.split(";")[0].split("//")[0]
                if VMEM.match(op):
                    if op.startswith("buffer_load") and re.search(r"\blds\b", code):
                        since_piece = 0          # a DMA piece of the NEXT stage: what the counted wait of the next barrier must cover
                        npieces += 1
                    else:
                        since_piece += 1
                m = WAIT.match(code)
                # the counted form (common.h: wait_barrier<N>) is ONE asm statement `s_waitcnt vmcnt(N) lgkmcnt(0); s_barrier`: in a listing
                # it sits between the ASM markers, in a disassembly the wait is the instruction directly in front of the barrier
                if in_asm and m and m.group(2):
                    last_wait = int(m.group(1))
                if op == "s_barrier":
                    nbar += 1
                    n = last_wait if in_asm else (prev_wait if from_objdump else None)
                    if n is not None:
                        nasm += 1
                        assert n <= since_piece, \
                            f"{name}: barrier #{nbar} lets {n} vector-memory instructions fly, only {since_piece} were emitted behind the stage's last DMA piece"
                        counted += n > 0
                prev_wait = int(m.group(1)) if (m and m.group(2)) else None
            assert npieces >= 8, (name, npieces)
            assert nbar >= min_barriers and nasm >= nbar // 2, (name, nbar, nasm)
            assert counted >= min_counted, f"{name}: only {counted} of {nbar} stage barriers carry a counted wait with N > 0"
            # no jump table left from advance_barrier_young's switch
            assert not any("s_setpc_b64" in ln for ln in body), f"{name}: indirect branch in the kernel body"


def test_stage_barriers_carry_counted_waits(isa):
    text, from_objdump = isa
    _check_counted_barriers(_bodies(text), KERNELS, from_objdump)


def test_stage_barriers_of_the_16_point_kernels(tmp_path):
    """The same invariant on mlp_x3h.o (in-tree object only: the 32-point fixture above covers the compile-from-source fallback)."""
    text = _disassemble_in_tree(str(tmp_path), OBJ_H, "mlp_x3h.hip")
    if text is None:
        pytest.skip("no fresh in-tree mlp_x3h.o: run __graft_entry__.build()")
    _check_counted_barriers(_bodies(text), KERNELS_H, True, min_barriers=8)


def test_stage_barriers_of_the_packing_instantiations(tmp_path):
    """The same invariant on mlp_x3p.o: the d sdf/dx chain and the sweeps with packed 24-bit records of abars / gus / gas (I2SDF_OPT_SAVES24; x3.h P24) -- a
    16-B + an 8-B access per tensor and k-chunk where the fp32 form has two 16-B ones, so the sources' instruction counts are unchanged.  In-tree object only."""
    text = _disassemble_in_tree(str(tmp_path), os.path.join(ROOT, "i2sdf_amd", "lib", "obj", "mlp_x3p.o"), "mlp_x3.hip")
    if text is None:
        pytest.skip("no fresh in-tree mlp_x3p.o: run __graft_entry__.build()")
    bodies = _bodies(text)
    assert not any("sdf_fwd3_kernel" in n for n in bodies), "mlp_x3p.o holds the packing instantiations only"
    _check_counted_barriers(bodies, KERNELS, True)
    for key in KERNELS:
        for m in re.finditer(r"\.name:\s+(\S*%s\S*)\n(?:.*\n){0,12}?\s+\.private_segment_fixed_size:\s+(\d+)" % key, text):
            assert int(m.group(2)) == 0, (m.group(1), m.group(2))


def test_no_scratch_in_the_counted_kernels(isa):
    isa = isa[0]
    for key in KERNELS:
        for m in re.finditer(r"\.name:\s+(\S*%s\S*)\n(?:.*\n){0,12}?\s+\.private_segment_fixed_size:\s+(\d+)" % key, isa):
            assert int(m.group(2)) == 0, (m.group(1), m.group(2))
        sizes = re.findall(r"\.private_segment_fixed_size:\s+(\d+)", isa)
        assert sizes, "no kernel metadata found"
