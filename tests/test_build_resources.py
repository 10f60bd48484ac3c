"""Register-budget regression guard for the ping-pong GEMM (CPU only: hipcc cross-compiles gfx950).

The 256x320 tile runs at 252-256 VGPRs; a careless edit pushes hundreds of registers to scratch and every layer shape
drops ~20x (seen during development).  This test compiles the three instantiation units to assembly with
-Rpass-analysis=kernel-resource-usage and checks, for every kernel: <= 256 VGPRs, two waves per SIMD, a bounded number
of spills, and -- what actually matters for speed -- that no scratch access sits in a hot block of the K loop: the
epilogue of the widest tile may park a few tile-invariant values in scratch once per output tile, the half-step
loop must not."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ctrlv_amd", "csrc")


def test_no_unguarded_store_data_hazard():
    """(Also: no MFMA result reaches an LDS store within 19 issue slots -- tools/hazard_scan.py scan_mfma_to_lds_store.)
    No buffer_store_dwordx4 with an SGPR soffset is followed within two wait states by a VALU write of its data
    registers -- the hazard LLVM exempts and gfx950 has (csrc/gemm_pp_kernel.h store_data_hazard_guard; the root cause of
    the round-3 "zero dwords" defect and of raw fp32 dwords in fp16 tiles) -- in any kernel of either element type."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import __graft_entry__ as g
    from hazard_scan import scan, scan_mfma_to_lds_store
    procs = []
    with tempfile.TemporaryDirectory() as td:
        for defs in ([], ["-DCTRLV_ELEM_F16=1"]):
            for unit in g.HIP_SOURCES:
                if unit == "abi.hip":
                    continue
                asm = os.path.join(td, unit + ("f16" if defs else "") + ".s")
                cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                       *g.EXTRA_FLAGS.get(unit, []), *defs, os.path.join(CSRC, unit), "-o", asm]
                procs.append((unit, asm, subprocess.Popen(cmd, stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)))
                if len(procs) % 8 == 0:
                    for _, _, p in procs[-8:]:
                        p.wait()
        n_stores = 0
        for unit, asm, p in procs:
            assert p.wait() == 0, unit
            text = open(asm).read()
            n_stores += text.count("buffer_store_dwordx4")
            hits = scan(text)
            assert not hits, (unit, hits[:3])
            # the asm staging stores of round 5 (stg_write16): no MFMA result is stored to LDS within the wait states LLVM
            # would have inserted for an instruction of its own
            near = scan_mfma_to_lds_store(text)
            assert not near, (unit, near[:3])
        assert n_stores > 500          # the scan really saw the epilogues


def test_pingpong_gemm_register_budget():
    procs = []
    with tempfile.TemporaryDirectory() as td:
        for unit in ("gemm_pp_m0.hip", "gemm_pp_m1.hip", "gemm_pp_m2.hip"):
            asm = os.path.join(td, unit + ".s")
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                   "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, unit), "-o", asm]
            procs.append((unit, asm, subprocess.Popen(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)))
        n_kernels = 0
        for unit, asm, p in procs:
            err = p.communicate()[1]
            assert p.returncode == 0, err[-2000:]
            vg = [int(x) for x in re.findall(r"VGPRs: (\d+)", err)]
            sp = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", err)]
            sc = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", err)]
            occ = [int(x) for x in re.findall(r"Occupancy \[waves/SIMD\]: (\d+)", err)]
            n_kernels += len(vg)
            assert vg and max(vg) <= 256, (unit, vg)
            assert all(o >= 2 for o in occ), (unit, occ)          # two waves per SIMD: the schedule depends on it
            assert max(sp) <= 64 and max(sc) <= 128, (unit, sp, sc)
            # scratch traffic inside the half-step loop (the blocks LLVM annotates "Depth=2"), per kernel
            text = open(asm).read()
            kernels = re.split(r"\n(?=_ZN\S*gemm_pp_kernel\S*:)", text)[1:]
            assert len(kernels) == len(vg), (unit, len(kernels), len(vg))
            for k in kernels:
                body = k.split("s_endpgm")[0].split("\n")
                assert any("v_mfma" in ln for ln in body)
                depth2 = [i for i, ln in enumerate(body) if "Depth=2" in ln]
                assert depth2, "K loop not found"
                # basic blocks of the K loop; the HOT ones are those every half-step executes (MFMAs, barriers, fragment
                # reads, LDS-DMA issue).  The once-per-tile `setup` branch (integer divisions for the tile coordinates)
                # may reload a few parked values; a hot block must not touch scratch.
                blocks, cur = [], []
                for ln in body[depth2[0]:depth2[-1] + 1]:
                    if ln.startswith(".LBB") or ln.startswith("; %bb."):
                        blocks.append(cur)
                        cur = []
                    cur.append(ln)
                blocks.append(cur)
                hot_marks = ("v_mfma", "s_barrier", "ds_read_b128", " lds")
                for blk in blocks:
                    if any(mk in ln for ln in blk for mk in hot_marks):
                        bad = [ln for ln in blk if "scratch_" in ln]
                        assert not bad, (unit, body[0][:90], bad)
        # 26 + two raw-output GEGLU variants + six 256x128 conv variants + nine row-halo 3x3 variants + four with producer-side
        # GroupNorm statistics (3x3 row-halo {V}, {R1}; temporal {V}, {R1}) + six K-slice variants (linear / 3x3 / temporal x 256 / 320 wide)
        assert n_kernels == 53


def test_temporal_fused_register_budget():
    """The fused temporal self-attention block (csrc/temporal_fused.hip) keeps a pixel's 32 x 320 input rows (80 registers) and
    the attention output (80) in registers beside the chains' accumulators: 240-256 VGPRs at two waves per SIMD.  The plain
    kernel must not spill at all (a spilled fragment is a scratch reload -- and a vmcnt(0) that drains the LDS-DMA ring -- inside
    a chain); the split-trunk instantiation of the fp16 library (two more residual / output planes in the epilogue) may park a
    few epilogue values."""
    procs = []
    with tempfile.TemporaryDirectory() as td:
        for defs in ([], ["-DCTRLV_ELEM_F16=1"]):
            asm = os.path.join(td, "tf" + ("16" if defs else "") + ".s")
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                   "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage", *defs, os.path.join(CSRC, "temporal_fused.hip"), "-o", asm]
            procs.append((defs, asm, subprocess.Popen(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)))
        for defs, asm, p in procs:
            err = p.communicate()[1]
            assert p.returncode == 0, err[-2000:]
            names = re.findall(r"Function Name: (\S+)", err)
            vg = [int(x) for x in re.findall(r" VGPRs: (\d+)", err)]
            sp = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", err)]
            occ = [int(x) for x in re.findall(r"Occupancy \[waves/SIMD\]: (\d+)", err)]
            kern = [(n, v, s, o) for n, v, s, o in zip(names, vg, sp, occ) if "temporal_fused_kernel" in n]
            assert len(kern) == (4 if defs else 2), (defs, names)        # <LO, LN>: plain / LayerNorm prologue (x split planes)
            for n, v, s, o in kern:
                assert v <= 256 and o >= 2, (n, v, o)
                assert s <= 8, (n, s)                  # (a few once-per-pixel-group values: strip address, next row offset)
            # ... and no scratch access inside a chain: no basic block with MFMAs of a chain touches scratch
            text = open(asm).read()
            for k in re.split(r"\n(?=_ZN\S*temporal_fused_kernel\S*:)", text)[1:]:
                blocks, cur = [], []
                for ln in k.split("s_endpgm")[0].split("\n"):
                    if ln.startswith(".LBB") or ln.startswith("; %bb."):
                        blocks.append(cur)
                        cur = []
                    cur.append(ln)
                blocks.append(cur)
                assert sum(ln.count("v_mfma") for b in blocks for ln in b) > 800
                for b in blocks:
                    if sum("v_mfma" in ln for ln in b) >= 8:
                        assert not [ln for ln in b if "scratch_" in ln], k[:80]


def test_wgrad_pp_register_budget():
    """The LDS-DMA weight-gradient kernel (csrc/wgrad_pp.hip): 160 accumulator registers + two fragment sets of 28 at two
    waves per SIMD (512 threads, 144 KiB of LDS: one workgroup per CU).  No spill in any of the three gather instantiations,
    for either element type, and the interleave of the loop survives: between two MFMAs of the K loop there are at most four
    transposed LDS reads (a long read-issue phase in front of the MFMAs is what the sched_barriers are there to prevent)."""
    procs = []
    with tempfile.TemporaryDirectory() as td:
        for defs in ([], ["-DCTRLV_ELEM_F16=1"]):
            asm = os.path.join(td, "wp" + ("16" if defs else "") + ".s")
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                   "-Rpass-analysis=kernel-resource-usage", *defs, os.path.join(CSRC, "wgrad_pp.hip"), "-o", asm]
            procs.append((defs, asm, subprocess.Popen(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)))
        for defs, asm, p in procs:
            err = p.communicate()[1]
            assert p.returncode == 0, err[-2000:]
            names = re.findall(r"Function Name: (\S+)", err)
            vg = [int(x) for x in re.findall(r" VGPRs: (\d+)", err)]
            sp = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", err)]
            scr = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", err)]
            occ = [int(x) for x in re.findall(r"Occupancy \[waves/SIMD\]: (\d+)", err)]
            kern = [(n, v, s, c, o) for n, v, s, c, o in zip(names, vg, sp, scr, occ) if "wgrad_pp_kernel" in n]
            assert len(kern) == 3, (defs, names)                         # linear / 3x3 / temporal gather
            for n, v, s, c, o in kern:
                assert v <= 256 and o >= 2 and s == 0 and c == 0, (n, v, s, c, o)
            text = open(asm).read()
            for k in re.split(r"\n(?=_ZN\S*wgrad_pp_kernel\S*:)", text)[1:]:
                body = k.split("s_endpgm")[0]
                assert body.count("v_mfma") == 20                        # one chunk: two sets of 5 x 2
                run, worst, seen = 0, 0, False
                for ln in body.split("\n"):
                    if "v_mfma" in ln:
                        seen, run = True, 0
                    elif "ds_read_b64_tr_b16" in ln and seen:
                        run += 1
                        worst = max(worst, run)
                    elif "s_barrier" in ln or "s_cbranch" in ln:
                        seen, run = False, 0
                assert 2 <= worst <= 4, worst
