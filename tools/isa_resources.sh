#!/bin/bash
# Static resource usage of every product kernel (no GPU needed): registers, scratch (spills), LDS and occupancy as the compiler
# reports them for gfx950 with the product flags.  -> profiles/rNN_isa_resources.txt
#   bash tools/isa_resources.sh 03
set -euo pipefail
R=${1:-03}
cd "$(dirname "$0")/.."
OUT=profiles/r${R}_isa_resources.txt
TMP=$(mktemp -d)
: > "$OUT"
printf "%-58s %5s %5s %7s %7s %4s\n" "kernel (source)" VGPR AGPR scratch LDS occ >> "$OUT"
# the product's own flags and per-source extras (vall-e-x_amd/_build.py is the one place they are written down)
flags_of() { python3 - "$1" <<'PY'
import importlib.util, os, sys
spec = importlib.util.spec_from_file_location("_vx_build", os.path.join("vall-e-x_amd", "_build.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
print(" ".join(m.FLAGS + m.EXTRA_FLAGS.get(sys.argv[1], [])))
PY
}
for f in vall-e-x_amd/csrc/*.hip; do
  # shellcheck disable=SC2046
  /opt/rocm/bin/hipcc $(flags_of "$(basename "$f")") -Rpass-analysis=kernel-resource-usage \
      -c "$f" -o "$TMP/x.o" 2> "$TMP/log" || { cat "$TMP/log"; exit 1; }
  python3 - "$TMP/log" "$(basename "$f")" >> "$OUT" <<'PY'
import re, sys, subprocess
log, src = sys.argv[1], sys.argv[2]
cur = None
rows = {}
for line in open(log):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    if cur is None:
        continue
    for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, line)
        if m:
            rows[cur][key] = m.group(1)
names = list(rows)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines() if names else []
for n, d in zip(names, dem):
    r = rows[n]
    d = re.sub(r"\(.*", "", d.replace("(anonymous namespace)::", "")).replace("void ", "").replace("vx::", "")
    print("%-58s %5s %5s %7s %7s %4s" % ((d + " (" + src.replace(".hip", "") + ")")[:58], r.get("vgpr", "?"), r.get("agpr", "?"),
                                          r.get("scratch", "?"), r.get("lds", "?"), r.get("occ", "?")))
PY
done
rm -rf "$TMP"
echo "wrote $OUT"
