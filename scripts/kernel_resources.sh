#!/bin/bash
# Per-kernel resource usage of a built library: scratch bytes per thread, spilled VGPRs, VGPRs, LDS (from the code object's notes).
# usage: scripts/kernel_resources.sh [isaac_aligner_amd/libisaac_gpu.so]
LIB=${1:-isaac_aligner_amd/libisaac_gpu.so}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input="$LIB" >/dev/null 2>&1
# the fat binary section holds the gfx950 code objects of every translation unit
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin="$TMP/fatbin" "$LIB" 2>/dev/null
python3 - "$TMP" <<'PY'
import sys, os, re, subprocess
tmp = sys.argv[1]
data = open(os.path.join(tmp, "fatbin"), "rb").read()
# code objects are ELF images inside the bundles
offs = [m.start() for m in re.finditer(b"\x7fELF", data)]
rows = []
for k, o in enumerate(offs):
    end = offs[k + 1] if k + 1 < len(offs) else len(data)
    p = os.path.join(tmp, "co%d.elf" % k)
    open(p, "wb").write(data[o:end])
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", p], capture_output=True, text=True).stdout
    cur = {}
    for line in out.splitlines():
        line = line.strip()
        m = re.match(r"-?\s*\.(\w+):\s*(.*)", line)
        if not m: continue
        key, val = m.group(1), m.group(2).strip()
        if key == "name" and "symbol" not in cur and val.startswith(("_Z", "k_")) is False: continue
        if key in ("name", "private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size", "symbol", "agpr_count"):
            cur[key] = val
        if key == "wavefront_size":
            if "symbol" in cur: rows.append(cur)
            cur = {}
seen = set()
print("%-44s %8s %6s %6s %6s %8s" % ("kernel", "scratch", "vgpr", "agpr", "vspill", "lds"))
for r in sorted(rows, key=lambda r: -int(r.get("private_segment_fixed_size", 0))):
    sym = r.get("symbol", "?").replace(".kd", "")
    name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
    if name in seen: continue
    seen.add(name)
    print("%-44s %8s %6s %6s %6s %8s" % (name[-44:], r.get("private_segment_fixed_size"), r.get("vgpr_count"), r.get("agpr_count", "0"), r.get("vgpr_spill_count"), r.get("group_segment_fixed_size")))
PY
