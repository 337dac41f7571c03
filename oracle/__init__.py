"""CPU oracle for the ImVoxelNet forward hot path -- TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this package.  The product (`imvoxelnet_amd/`) never does; it fails
loudly when the HIP library is missing instead of falling back to anything here.

Contents
  ivx_oracle.c        plain-C restatement (projection, unprojection, direct conv,
                      rotated-BEV IoU + NMS, aligned 3-D NMS); built by `make`
  c_oracle.py         ctypes bindings for the above
  imvoxel_oracle.py   numpy / torch-fp32-CPU restatement of the whole path
                      (anchors, coder, heads, necks, ResNet-50, FPN, detector)
  ref_import.py       imports the REAL reference from /root/reference (build
                      container only) -- used to generate tests/golden/
  gen_golden.py       the script that generated tests/golden/*.npz
  cpu_abi/            the SAME C-ABI as the product (include/imvoxel.h) served by a CPU restatement: cpu_ops.cpp
                      (op-level entry points on host memory) under the product's csrc/model.cpp compiled unchanged
                      against a host-memory stand-in of the HIP runtime (cpu_abi/hip/); build.py ->
                      oracle/_cpuabi/libimvoxel_cpu.so + tests/c/e2e_small_cpu.  Loaded by tests/ only.

Pinning status (see DESIGN.md "Oracle"):
  pinned   : get_points, _compute_projection, backproject(+mean), 3-D necks,
             Anchor3DHead decode, anchor generator, box coder, box utils,
             aligned_3d_nms, box3d_multiclass_nms  (golden vectors from the
             imported reference + the reference's own test vectors)
  partial  : rotated-BEV overlap (reference test_box3d.py known answers,
             rtol 1e-4); nms_gpu greedy scan (restated, CUDA op cannot run here)
  UNPINNED : ResNet-50 / FPN (mmdet 2.10.0 + torchvision, absent from the
             reference tree and this image) -- restated from the public
             architecture; DCNv2 (mmcv-full 1.2.7) not built.
"""
