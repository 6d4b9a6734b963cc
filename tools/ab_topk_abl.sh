#!/bin/bash
# ablation builds of the retrieval kernels (timing experiments, wrong results): libasr_hip_abl<N>.so next to the library
# ASR_TF_ABL bit 1: the filter epilogue never triggers; bit 2: no MFMAs
set -e
cd "$(dirname "$0")/../audio_sheet_retrieval_amd"
python -m audio_sheet_retrieval_amd.build >/dev/null 2>&1 || (cd .. && python -m audio_sheet_retrieval_amd.build >/dev/null)
for abl in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Wno-unused-result -Wno-unused-value -I ../include -DASR_TF_ABL=$abl -c csrc/tail_rank_kernels.hip -o /tmp/trk_abl$abl.o
  objs=$(ls csrc/_obj/*.o | grep -v tail_rank_kernels.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libasr_hip_abl$abl.so $objs /tmp/trk_abl$abl.o
  echo built libasr_hip_abl$abl.so
done
