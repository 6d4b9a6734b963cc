#!/bin/bash
# Build a VARIANT of the library HERE (hipcc cross-compiles): the listed source files recompiled with extra -D flags,
# every other object taken from the default build.  The result, audio_sheet_retrieval_amd/libasr_hip_<name>.so, travels
# to the GPU box with the snapshot; experiments load it through ASR_LIB_PATH (tools/ab_topk.py, tools/train_batch_sizes.py -
# the product loads libasr_hip.so and checks its source hash).
#   usage: tools/build_variant.sh <name> "<file1.hip file2.hip ...>" "<flags>"
set -e
cd "$(dirname "$0")/.."
python -m audio_sheet_retrieval_amd.build > /dev/null 2>&1
name=$1; srcs=$2; flags=$3
objs=$(ls audio_sheet_retrieval_amd/csrc/_obj/*.o)
for src in $srcs; do
  base=$(basename $src .hip)
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Wno-unused-result -Wno-unused-value -I include $flags \
      -c audio_sheet_retrieval_amd/csrc/$base.hip -o /tmp/${base}_$name.o 2> /tmp/${base}_$name.log ) &
  objs=$(echo "$objs" | grep -v "/$base.o"; echo /tmp/${base}_$name.o)
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o audio_sheet_retrieval_amd/libasr_hip_$name.so $objs
echo built audio_sheet_retrieval_amd/libasr_hip_$name.so
