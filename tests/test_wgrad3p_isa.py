"""CPU (hipcc cross-compiles gfx950 without a GPU): the contract between wgrad3p's inline asm and the compiler, checked on the ISA.

`wgrad3p_kernel` keeps its sixteen accumulator tiles in a[0:255] as state the compiler does not see (i2sdf_amd/csrc/wgrad.hip,
W3_ASM_MFMA: inline-asm zero fill / MFMA / read-out that name the registers themselves, every statement clobbering all AGPRs).  That is
only sound -- and only pays -- while the compiler
  1. references no AGPR of its own inside those kernels (it would overwrite, or be overwritten by, a tile),
  2. spills nothing to scratch there (a spill's reload waits with vmcnt(0) for the operand loads in flight), and
  3. does not copy in-flight rows between registers in the stage loops (v_mov behind a low vmcnt: the load-to-use distance collapses;
     this is what a one-stage-per-iteration loop compiled to, DESIGN.md "wgrad3p under the microscope").
A compiler update or an edit of the stage can break any of the three without failing a numerics test on small inputs."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "i2sdf_amd", "csrc")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    out = tmp_path_factory.mktemp("isa") / "wgrad.s"
    # the flags of i2sdf_amd/csrc/build.sh for wgrad.hip
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-Wno-unused-result",
           "-mllvm", "-pragma-unroll-threshold=1000000", "-fno-slp-vectorize", "-x", "hip", "--cuda-device-only", "-S",
           os.path.join(CSRC, "wgrad.hip"), "-o", str(out), "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read().splitlines(), r.stderr


def _kernels(lines):
    """{mangled name: lines} of the wgrad3p kernels"""
    out, cur = {}, None
    for ln in lines:
        m = re.match(r"^(_Z\w*wgrad3p_kernel\w*):", ln)
        if m:
            cur = m.group(1); out[cur] = []
        elif cur is not None:
            if ln.startswith(".Lfunc_end"):
                cur = None
            else:
                out[cur].append(ln)
    return out


def test_wgrad3p_accumulator_is_hidden_state(isa):
    lines, remarks = isa
    ks = _kernels(lines)
    assert len(ks) == 2, list(ks)                       # bf16x3 and bf16x2
    for name, body in ks.items():
        in_asm, n_mfma, foreign = False, 0, []
        for ln in body:
            if "#ASMSTART" in ln:
                in_asm = True
            elif "#ASMEND" in ln:
                in_asm = False
            code = ln.split(";")[0]
            if "v_mfma" in code:
                n_mfma += 1
                assert in_asm, f"{name}: an MFMA outside the inline asm: {ln.strip()}"
            if not in_asm and re.search(r"[ ,]a(\[\d+|\d+)", code) and not code.lstrip().startswith("."):
                foreign.append(ln.strip())
        assert n_mfma >= 8 * 96, (name, n_mfma)          # eight specialisations of the body, at least one stage each
        assert not foreign, f"{name}: the compiler uses AGPRs of its own: {foreign[:5]}"
    # resource remarks: no scratch, all 256 AGPRs accounted to the kernels (the clobber lists)
    for m in re.finditer(r"Function Name: (\S*wgrad3p_kernel\S*)(.*?)Occupancy", remarks, re.S):
        blk = m.group(2)
        assert re.search(r"ScratchSize \[bytes/lane\]: 0\b", blk), (m.group(1), blk)
        assert re.search(r"AGPRs: 256\b", blk), (m.group(1), blk)


def test_wgrad3p_stage_loops_do_not_copy_rows_in_flight(isa):
    lines, _ = isa
    for name, body in _kernels(lines).items():
        # basic blocks that branch back to themselves and carry MFMAs = the stage loops
        blocks, cur = {}, None
        for ln in body:
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                cur = m.group(1); blocks[cur] = []
            elif cur is not None:
                blocks[cur].append(ln.split(";")[0])
        loops = {k: v for k, v in blocks.items() if sum("v_mfma" in x for x in v) >= 96 and any(re.search(r"s_cbranch\w+ " + re.escape(k) + r"\b", x) for x in v)}
        assert loops, name
        for k, v in loops.items():
            assert not any("scratch_" in x for x in v), (name, k)
            assert not any(re.search(r"\bv_mov_b(32|64)", x) for x in v), (name, k, [x.strip() for x in v if "v_mov_b" in x][:4])
            assert not any(re.search(r"v_accvgpr", x) for x in v), (name, k)
            low = [x.strip() for x in v if re.search(r"s_waitcnt vmcnt\([0-3]\)", x)]
            assert not low, (name, k, low)               # eight row loads per stage in flight: the waits are vmcnt(5) ... vmcnt(7)


def test_wgrad_narrow_bf16_accumulator_window(isa):
    """wgrad_narrow_kernel<3> (bf16 split form of the narrow blocks, round 4; always three planes) keeps its accumulator tiles in a[...] as hidden state too,
    but only DURING the accumulation: zero fill -> inline-asm MFMAs -> read-out, after which the cross-wave reduction is ordinary code in
    which the compiler may use AGPRs again.  Inside that window -- per instantiated body -- there must be no compiler-owned AGPR reference
    (it would overwrite a tile: the clobber lists only say that the asm destroys the registers, not that they carry state between the
    statements), and every MFMA must be one of the asm statements."""
    lines, _ = isa
    ks, cur = {}, None
    for ln in lines:
        m = re.match(r"^(_Z\w*wgrad_narrow_kernelILi3E\w*):", ln)
        if m:
            cur = m.group(1); ks[cur] = []
        elif cur is not None:
            if ln.startswith(".Lfunc_end"):
                cur = None
            else:
                ks[cur].append(ln)
    assert len(ks) == 1, list(ks)
    for name, body in ks.items():
        in_asm, events = False, []                      # (kind, text): Z zero fill, M asm MFMA, R asm read-out, C compiler AGPR reference
        for ln in body:
            if "#ASMSTART" in ln:
                in_asm = True; continue
            if "#ASMEND" in ln:
                in_asm = False; continue
            code = ln.split(";")[0].strip()
            if not code or code.startswith("."):
                continue
            if in_asm:
                if code.startswith("v_mfma"):
                    events.append(("M", code))
                elif code.startswith("v_accvgpr_read"):
                    events.append(("R", code))
                elif code.startswith("v_accvgpr_write") or code.startswith(".rept"):
                    events.append(("Z", code))
            else:
                assert "v_mfma" not in code, f"{name}: an MFMA outside the inline asm: {code}"
                if re.search(r"[ ,]a(\[\d+|\d+)", " " + code):
                    events.append(("C", code))
        kinds = "".join(k for k, _ in events)
        runs = re.sub(r"(.)\1+", r"\1", kinds)          # e.g. ZMRCZMRCZMRC: three operand-shape variants of the body
        assert re.fullmatch(r"(ZMRC?)+", runs), f"{name}: unexpected order of zero fill / MFMAs / read-out / compiler AGPR uses: {runs}"
        assert kinds.count("M") >= 3 * 12, (name, kinds.count("M"))


def test_wgrad_narrow_bf16_has_no_spills_in_its_loops(isa):
    """Round 5 (VERDICT r4 task 5a): the shipped round-4 binary of wgrad_narrow_kernel<3> carried 108 B of scratch per lane (26 spilled VGPRs: the
    4 x 2-tile variant read all eight tiles out of a[...] into VGPRs in front of its cross-wave sum and store loop).  The tiles now leave the
    AGPRs one at a time; what may remain is a single value parked at kernel entry (<= 8 B), never a spill or reload inside a loop."""
    lines, remarks = isa
    m = re.search(r"Function Name: (\S*wgrad_narrow_kernelILi3E\S*)(.*?)Occupancy", remarks, re.S)
    assert m, "no resource remark for wgrad_narrow_kernel<3>"
    sz = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", m.group(2)).group(1))
    assert sz <= 8, sz
    # blocks that belong to a loop carry the assembler's "in Loop:" / "Loop Header" annotation on their label line
    cur, in_loop, bad, n_loop_blocks = False, False, [], 0
    for ln in lines:
        if re.match(r"^_Z\w*wgrad_narrow_kernelILi3E\w*:", ln):
            cur = True
        elif cur and ln.startswith(".Lfunc_end"):
            break
        elif cur:
            if re.match(r"^\.LBB\d+_\d+:", ln):
                in_loop = "Loop" in ln
                n_loop_blocks += in_loop
            elif in_loop and "scratch_" in ln.split(";")[0]:
                bad.append(ln.strip())
    assert n_loop_blocks > 10
    assert not bad, bad[:5]


# ---- round 6: wgrad_all_kernel = the 256x256 blocks AND the narrow / 128x128 tasks of a point range in ONE grid (the kernel the default
# path launches); the same three contracts on the merged kernel, where one body's register needs could leak into another's code
def _bodies_named(lines, pat):
    out, cur = {}, None
    for ln in lines:
        m = re.match(r"^(_Z\w*" + pat + r"\w*):", ln)
        if m:
            cur = m.group(1); out[cur] = []
        elif cur is not None:
            if ln.startswith(".Lfunc_end"):
                cur = None
            else:
                out[cur].append(ln)
    return out


def test_wgrad_all_kernel_keeps_the_accumulator_windows(isa):
    lines, remarks = isa
    ks = _bodies_named(lines, "wgrad_all_kernel")
    assert len(ks) == 2, list(ks)                       # two / three planes for the 256x256 blocks
    for name, body in ks.items():
        in_asm, events = False, []
        for ln in body:
            if "#ASMSTART" in ln:
                in_asm = True; continue
            if "#ASMEND" in ln:
                in_asm = False; continue
            code = ln.split(";")[0].strip()
            if not code or code.startswith("."):
                continue
            if in_asm:
                if code.startswith("v_mfma"):
                    events.append("M")
                elif code.startswith("v_accvgpr_read"):
                    events.append("R")
                elif code.startswith("v_accvgpr_write") or code.startswith(".rept"):
                    events.append("Z")
            else:
                assert "v_mfma" not in code, f"{name}: an MFMA outside the inline asm: {code}"
                if re.search(r"[ ,]a(\[\d+|\d+)", " " + code):
                    events.append("C")
        runs = re.sub(r"(.)\1+", r"\1", "".join(events))
        # every body: zero fill -> asm MFMAs -> read-out [-> ordinary code that may use AGPRs: only the narrow bodies' cross-wave sums]
        assert re.fullmatch(r"(ZMRC?)+", runs), f"{name}: unexpected order of zero fill / MFMAs / read-out / compiler AGPR uses: {runs}"
        assert events.count("M") >= 8 * 48 + 3 * 12, (name, events.count("M"))
    for m in re.finditer(r"Function Name: (\S*wgrad_all_kernel\S*)(.*?)Occupancy", remarks, re.S):
        sz = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", m.group(2)).group(1))
        assert sz <= 8, (m.group(1), sz)


def test_wgrad_all_kernel_loops_are_clean(isa):
    """the stage loops of the 256x256 bodies inside the merged kernel: no scratch, no row copies, no low vmcnt waits (as for wgrad3p_kernel
    above), and no scratch access inside ANY loop of the kernel"""
    lines, _ = isa
    for name, body in _bodies_named(lines, "wgrad_all_kernel").items():
        blocks, cur, loop_blocks, bad = {}, None, 0, []
        in_loop = False
        for ln in body:
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                cur = m.group(1); blocks[cur] = []
                in_loop = "Loop" in ln
                loop_blocks += in_loop
            elif cur is not None:
                blocks[cur].append(ln.split(";")[0])
                if in_loop and "scratch_" in ln.split(";")[0]:
                    bad.append(ln.strip())
        assert loop_blocks > 10 and not bad, (name, bad[:5])
        loops = {k: v for k, v in blocks.items() if sum("v_mfma" in x for x in v) >= 96 and any(re.search(r"s_cbranch\w+ " + re.escape(k) + r"\b", x) for x in v)}
        assert loops, name
        for k, v in loops.items():
            assert not any(re.search(r"\bv_mov_b(32|64)", x) for x in v), (name, k, [x.strip() for x in v if "v_mov_b" in x][:4])
            assert not any(re.search(r"v_accvgpr", x) for x in v), (name, k)
            low = [x.strip() for x in v if re.search(r"s_waitcnt vmcnt\([0-3]\)", x)]
            assert not low, (name, k, low)
