#!/bin/bash
# Debug build of the library for tools/conv_timeline.py / tools/conv_timeline_model.py / tools/halo_timeline.py: conv_igemm.hip's host TU (0),
# fp16-pair TU (4) and z-halo TU (5) with -DIVX_CONV_TIMELINE (per-workgroup s_memrealtime stamps in conv_igemm_v4_kernel's pair-IO path and in
# conv_wino_halo_kernel; ivx_conv_set_timeline) and bottleneck.hip (ivx_bottleneck_set_timeline: tools/bottleneck_timeline.py), linked with the product build's other objects into tools/bin/libimvoxel_hip_tl.so
# (git-ignored; ~50 MB).  Run `python -m imvoxelnet_amd._build` first.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/imvoxelnet_amd/csrc
TMP=${TMPDIR:-/tmp}/ivx_tl
mkdir -p "$TMP" "$ROOT/tools/bin"
for tu in 0 4 5; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$CSRC/conv_igemm.hip" -o "$TMP/tu$tu.o" -DIVX_CONV_TU=$tu -DIVX_CONV_TIMELINE &
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$CSRC/bottleneck.hip" -o "$TMP/bottleneck.o" -DIVX_CONV_TIMELINE &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$CSRC/stem.hip" -o "$TMP/stem.o" -DIVX_CONV_TIMELINE &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$CSRC/winograd.hip" -o "$TMP/winograd.o" -DIVX_CONV_TIMELINE &
wait
cd "$CSRC"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/bin/libimvoxel_hip_tl.so" "$TMP/tu0.o" conv_igemm_f32.o conv_igemm_lowp.o \
  conv_igemm_pair_bf16.o "$TMP/tu4.o" "$TMP/tu5.o" "$TMP/bottleneck.o" "$TMP/stem.o" "$TMP/winograd.o" pool_layout.o backproject.o anchor_tail.o dcn.o fold4w.o ubench.o api_common.o kitti_eval.o model.o
echo "$ROOT/tools/bin/libimvoxel_hip_tl.so"
