#!/bin/bash
# HERE (no GPU needed): compile every csrc/*.hip to a device listing and print the kernels / device functions that use
# scratch memory, with the loop depth of every scratch instruction.  Spill stores in a prologue are harmless; a reload
# inside a loop (depth >= 1) followed by s_waitcnt vmcnt(0) is a memory round trip per iteration - round 6 found the
# training step's Jacobi rounds (34 per round) and the RAW Winograd builds that way.
#   usage: tools/scratch_scan.sh [file.hip ...]      (default: all of csrc/)
cd "$(dirname "$0")/.."
OUT=${TMPDIR:-/tmp}/asr_isa; mkdir -p $OUT
FILES=${@:-$(ls audio_sheet_retrieval_amd/csrc/*.hip | grep -v asr_version)}
for f in $FILES; do
  b=$(basename $f .hip)
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -S --cuda-device-only -Wno-comment -Wno-unused-result \
      -Wno-unused-value $f -o $OUT/$b.s 2> /dev/null ) &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.5; done
done
wait
python3 - $OUT <<'PY'
import glob, os, re, sys
for path in sorted(glob.glob(os.path.join(sys.argv[1], "*.s"))):
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z[\w$.]+):", lines[i])
        if not m:
            i += 1
            continue
        j = i + 1
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        tail = lines[j:j + 80]
        sc = next((int(re.search(r"(\d+)", l).group(1)) for l in tail if "; ScratchSize:" in l), 0)
        if sc:
            vg = next((l.split(":")[1].strip() for l in tail if "; NumVgprs:" in l), "?")
            depth, inloop, waits = 0, 0, 0
            body = lines[i:j]
            for k, l in enumerate(body):
                if l.startswith(".LBB"):
                    mm = re.search(r"Depth=(\d+)", l)
                    depth = int(mm.group(1)) if mm else 0
                if "scratch_load" in l and depth >= 1:
                    inloop += 1
                    if any("s_waitcnt vmcnt(0)" in x for x in body[k + 1:k + 3]):
                        waits += 1
            print("%-22s %-96s scratch %4d B  vgprs %s  reloads inside loops %3d (waited at once: %d)"
                  % (os.path.basename(path)[:-2], m.group(1)[:96], sc, vg, inloop, waits))
        i = j
PY
