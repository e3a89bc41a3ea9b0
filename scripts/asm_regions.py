#!/usr/bin/env python3
"""Developer tool: static instruction counts per PE_MARK region of one kernel (no GPU needed).
   python scripts/asm_regions.py [kernel-substring]   (compiles pe_kernels.hip with -DPE_ASM_MARKS to /tmp/k_marks.s)"""
import collections, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "phy-engine_amd", "csrc")
out = "/tmp/k_marks.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-I../include", "-std=c++20", "-O3", "-fPIC", "-DNDEBUG", "-DPE_ASM_MARKS", "--cuda-device-only",
                "-S", "-o", out, "-x", "hip", "pe_kernels.hip"], cwd=src, check=True, stderr=subprocess.DEVNULL)
want = sys.argv[1] if len(sys.argv) > 1 else "k_m2_factor_partsILi4"
lines = open(out).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l and l.rstrip().endswith(":") is False and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
region, counts = "head", collections.OrderedDict()
for l in lines[start:end]:
    m = re.search(r"; PE_MARK (\w+)", l)
    if m:
        region = m.group(1) + "#" + str(sum(1 for k in counts if k.split("#")[0] == m.group(1)))
        continue
    t = l.strip().split()
    if not t or not re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", t[0]):
        continue
    c = counts.setdefault(region, collections.Counter())
    op = t[0]
    kind = ("mfma" if "mfma" in op else "readlane" if "readlane" in op or "writelane" in op else "valu" if op.startswith("v_") else
            "branch" if op.startswith("s_cbranch") or op == "s_branch" else "waitcnt" if op == "s_waitcnt" else "salu" if op.startswith("s_") else
            "lds" if op.startswith("ds_") else "vmem")
    c[kind] += 1
    c["total"] += 1
for r, c in counts.items():
    print(f"{r:14s} " + " ".join(f"{k}={c[k]}" for k in ("total", "valu", "readlane", "mfma", "salu", "branch", "waitcnt", "lds", "vmem")))
