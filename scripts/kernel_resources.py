#!/usr/bin/env python
"""Static resource table of every gfx950 kernel in librfx.so, read from the code objects inside the library (no GPU needed).

    python scripts/kernel_resources.py [--tsv profiles/rNN_kernel_resources.tsv]

The library's ``.hip_fatbin`` section holds one offload bundle per translation unit (in link order: the Makefile's SRCS without the
kernel-less api.hip).  Each bundle's gfx950 code object is taken out (llvm-objcopy + clang-offload-bundler) and its amdhsa metadata
note parsed: registers (arch VGPRs + accumulation VGPRs, SGPRs), LDS bytes, scratch bytes, spill counts, the workgroup size the
kernel was compiled for.  From those, the occupancy the hardware grants (MI355X_MICROARCH.md: 512 VGPRs per SIMD lane in the
unified arch+acc file, allocated in blocks of 8; 160 KB LDS per CU; 4 SIMDs per CU; at most 8 wavefronts per SIMD): wavefronts
per SIMD and workgroups per CU.

For a kernel that touches scratch the disassembly says WHERE: ``k_loop_scratch`` counts the scratch instructions inside the
innermost loops that issue MFMAs (the K loops: a loop = a backward branch and its target; innermost = no other MFMA loop nested in
it; for a kernel without MFMAs, the loops with at least 16 FMAs), ``k_loop_mfma`` the MFMAs (FMAs) of the largest such loop, ``other_scratch`` the scratch instructions everywhere else (prologue, the
per-tile set-up of a persistent kernel, the epilogue).  A spill outside the K loops costs a few memory instructions per tile; one
inside is paid per K step.

tests/test_oracle.py holds the table against what DESIGN.md says about the hot kernels.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ransac-flow_amd", "csrc")
LIB = os.path.join(ROOT, "ransac-flow_amd", "librfx.so")
LLVM = os.environ.get("RFX_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
VGPR_FILE, VGPR_BLOCK, LDS_BYTES, SIMDS, MAX_WAVES = 512, 8, 160 * 1024, 4, 8
COLS = ["unit", "kernel", "wg", "vgpr", "agpr", "sgpr", "lds", "scratch", "vgpr_spills", "sgpr_spills", "waves_per_simd", "wg_per_cu",
        "k_loop_mfma", "k_loop_scratch", "other_scratch"]


def tool(name):
    return os.path.join(LLVM, name)


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    res = []
    for l in out.stdout.splitlines():
        l = l.replace("(anonymous namespace)::", "").replace("void ", "", 1)
        depth = 0
        for i, ch in enumerate(l):                                   # the kernel's name with its template arguments, without the
            depth += ch == "<"                                       # parameter list
            depth -= ch == ">"
            if ch == "(" and depth == 0:
                l = l[:i]
                break
        res.append(l)
    return res


def unit_names(n):
    """Labels of the n bundles: the Makefile's link order, minus the units without device code."""
    m = re.search(r"^SRCS\s*=\s*(.*)$", open(os.path.join(CSRC, "Makefile")).read(), flags=re.M)
    srcs = [s[:-4] for s in m.group(1).split()] if m else []
    with_kernels = [s for s in srcs if "__global__" in open(os.path.join(CSRC, s + ".hip")).read()]
    return with_kernels if len(with_kernels) == n else ["unit%d" % i for i in range(n)]


def code_objects(lib, workdir):
    """Paths of the gfx950 code objects of the library, one per translation unit, in link order."""
    fb = os.path.join(workdir, "fatbin")
    subprocess.run([tool("llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, lib, os.path.join(workdir, "copy")],
                   check=True, capture_output=True)
    data = open(fb, "rb").read()
    starts = [m.start() for m in re.finditer(MAGIC, data)] + [len(data)]
    cos = []
    for i in range(len(starts) - 1):
        bundle, co = os.path.join(workdir, "b%d" % i), os.path.join(workdir, "co%d" % i)
        with open(bundle, "wb") as f:
            f.write(data[starts[i]:starts[i + 1]])
        subprocess.run([tool("clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + bundle, "--targets=" + TARGET,
                        "--output=" + co], check=True, capture_output=True)
        cos.append(co)
    return cos


def metadata(co):
    """[dict of the amdhsa.kernels fields] of one code object."""
    notes = subprocess.run([tool("llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels, cur, inside = [], None, False
    for line in notes.splitlines():
        if line.startswith("amdhsa.kernels:"):
            inside = True
            continue
        if inside and re.match(r"^amdhsa\.\w+:", line):
            inside = False
        m = inside and re.match(r"^  (-| ) \.(\w+):\s*(.*)$", line)  # top-level keys of one kernel entry (.args nest deeper)
        if not m:
            continue
        if m.group(1) == "-":
            cur = {}
            kernels.append(cur)
        cur[m.group(2)] = m.group(3).strip()
    return [k for k in kernels if "name" in k]


def k_loops(co, mangled):
    """(instructions [(offset, opcode)], the kernel's K loops [(lo, hi)], offsets of its MFMAs -- FMAs for a kernel without MFMAs)."""
    asm = subprocess.run([tool("llvm-objdump"), "-d", "--disassemble-symbols=" + mangled, co], check=True, capture_output=True,
                         text=True).stdout
    ins, base = [], None
    for l in asm.splitlines():
        m = re.match(r"^\t(\S+)\s.*// ([0-9A-F]+):", l)
        if not m:
            continue
        addr = int(m.group(2), 16)
        base = addr if base is None else base
        tgt = re.search(r"\+0x([0-9a-f]+)>\s*$", l) if m.group(1).startswith(("s_cbranch", "s_branch")) else None
        ins.append((addr - base, m.group(1), int(tgt.group(1), 16) if tgt else None))
    loops = [(t, a) for a, op, t in ins if t is not None and t <= a]
    mf = [a for a, op, _ in ins if op.startswith("v_mfma")]
    need = 1
    if not mf:                                                       # a kernel without MFMAs (the 7x7 correlation): its FMA loops
        mf, need = [a for a, op, _ in ins if op.startswith(("v_fmac_f32", "v_pk_fma_f32", "v_fma_f32"))], 16
    mfma_loops = [(lo, hi) for lo, hi in loops if sum(lo <= a <= hi for a in mf) >= need]
    inner = [(lo, hi) for lo, hi in mfma_loops
             if not any((l2, h2) != (lo, hi) and lo <= l2 and h2 <= hi for l2, h2 in mfma_loops)]
    return [(a, op) for a, op, _ in ins], inner, mf


def scratch_placement(co, mangled):
    """(MFMAs of the largest K loop, scratch instructions inside K loops, scratch instructions elsewhere) of one kernel."""
    ins, inner, mf = k_loops(co, mangled)
    sc = [a for a, op in ins if op.startswith("scratch_")]
    hot = sum(1 for a in sc if any(lo <= a <= hi for lo, hi in inner))
    biggest = max([sum(1 for a in mf if lo <= a <= hi) for lo, hi in inner] or [0])
    return biggest, hot, len(sc) - hot


MIX = ["mfma", "ds_read", "ds_write", "vmem_load", "vmem_store", "scratch", "valu", "salu", "waitcnt", "barrier", "other"]


def instruction_class(op):
    for prefix, c in (("v_mfma", "mfma"), ("ds_read", "ds_read"), ("ds_write", "ds_write"), ("global_load", "vmem_load"),
                      ("buffer_load", "vmem_load"), ("global_store", "vmem_store"), ("buffer_store", "vmem_store"),
                      ("scratch_", "scratch"), ("v_", "valu"), ("s_waitcnt", "waitcnt"), ("s_barrier", "barrier"), ("s_", "salu")):
        if op.startswith(prefix):
            return c
    return "other"


def k_loop_mix(co, mangled):
    """Instruction counts by class of the kernel's largest K loop (one iteration = one K step of the tile)."""
    ins, inner, mf = k_loops(co, mangled)
    if not inner:
        return None
    lo, hi = max(inner, key=lambda r: sum(r[0] <= a <= r[1] for a in mf))
    c = collections.Counter(instruction_class(op) for a, op in ins if lo <= a <= hi)
    return dict({k: c.get(k, 0) for k in MIX}, bytes=hi - lo)


def mix_table(pattern, lib=LIB):
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for co in code_objects(lib, d):
            ks = metadata(co)
            for k, name in zip(ks, demangle([k["name"] for k in ks])):
                if re.search(pattern, name):
                    m = k_loop_mix(co, k["name"])
                    if m:
                        rows.append(dict(m, kernel=name))
    return rows


def occupancy(vgpr, agpr, lds, wg):
    regs = -(-max(vgpr + agpr, 1) // VGPR_BLOCK) * VGPR_BLOCK       # unified file: arch + acc, rounded to the allocation block
    waves_simd = min(MAX_WAVES, VGPR_FILE // regs)
    waves_wg = -(-wg // 64)
    by_regs = waves_simd * SIMDS // waves_wg
    by_lds = LDS_BYTES // lds if lds else 1 << 30
    return waves_simd, min(by_regs, by_lds, MAX_WAVES * SIMDS // waves_wg)


def table(lib=LIB):
    rows = []
    with tempfile.TemporaryDirectory() as d:
        cos = code_objects(lib, d)
        for unit, co in zip(unit_names(len(cos)), cos):
            ks = metadata(co)
            for k, name in zip(ks, demangle([k["name"] for k in ks])):
                g = lambda f: int(k.get(f, 0))
                ws, wc = occupancy(g("vgpr_count"), g("agpr_count"), g("group_segment_fixed_size"), g("max_flat_workgroup_size"))
                place = scratch_placement(co, k["name"]) if g("private_segment_fixed_size") else (0, 0, 0)
                rows.append(dict(unit=unit, kernel=name, wg=g("max_flat_workgroup_size"), vgpr=g("vgpr_count"), agpr=g("agpr_count"),
                                 sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size"),
                                 vgpr_spills=g("vgpr_spill_count"), sgpr_spills=g("sgpr_spill_count"), waves_per_simd=ws,
                                 wg_per_cu=wc, k_loop_mfma=place[0], k_loop_scratch=place[1], other_scratch=place[2]))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tsv")
    ap.add_argument("--lib", default=LIB)
    ap.add_argument("--mix", metavar="REGEX", help="instead: the instruction mix of the largest K loop of the kernels matching REGEX")
    a = ap.parse_args()
    if a.mix:
        cols = ["kernel", "bytes"] + MIX
        text = "\t".join(cols) + "\n" + "".join("\t".join(str(r[c]) for c in cols) + "\n" for r in mix_table(a.mix, a.lib))
        if a.tsv:
            with open(a.tsv, "w") as f:
                f.write("# instruction mix of one iteration of the largest K loop (scripts/kernel_resources.py --mix; static, from"
                        " librfx.so's gfx950 code objects)\n" + text)
        sys.stdout.write(text)
        return
    rows = table(a.lib)
    text = "\t".join(COLS) + "\n" + "".join("\t".join(str(r[c]) for c in COLS) + "\n" for r in rows)
    if a.tsv:
        with open(a.tsv, "w") as f:
            f.write("# static kernel resources of librfx.so (scripts/kernel_resources.py: hipcc --offload-arch=gfx950 -O3 code objects,"
                    " no GPU involved); k_loop_* = inside the innermost MFMA loops\n" + text)
    sys.stdout.write(text)
    n = collections.Counter(r["scratch"] > 0 for r in rows)
    sys.stderr.write("%d kernels, %d touch scratch, %d of those inside a K loop\n"
                     % (len(rows), n[True], sum(r["k_loop_scratch"] > 0 for r in rows)))


if __name__ == "__main__":
    main()
