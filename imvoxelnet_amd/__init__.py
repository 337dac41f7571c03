"""imvoxelnet_amd -- MI355X-native ImVoxelNet forward path (hand-written HIP for gfx950 behind a C-ABI).

Importing the package registers the modules under the reference's registry names in its OWN registries (aliasing them
into mmdet's is an explicit register_into_mmdet() call or IVX_REGISTER_MMDET=1); it does NOT load the
HIP library (that happens on first use and fails loudly if libimvoxel_hip.so is missing).
"""
from .registry import (BACKBONES, NECKS, HEADS, DETECTORS, ANCHOR_GENERATORS, BBOX_CODERS, ConfigDict,  # noqa: F401
                       build_backbone, build_neck, build_head, build_detector)
from .backbones import ResNet, FPN                                           # noqa: F401
from .necks3d import (KittiImVoxelNeck, NuScenesImVoxelNeck, FastIndoorImVoxelNeck, ImVoxelNeck, BasicBlock3d,  # noqa: F401
                      BasicBlock3dV2)
from .anchor import Anchor3DRangeGenerator, DeltaXYZWLHRBBoxCoder          # noqa: F401
from .heads import Anchor3DHead                                            # noqa: F401
from .heads_layout import LayoutHead, get_extrinsics                       # noqa: F401
from .heads_indoor import (ScanNetImVoxelHeadV2, SunRgbdImVoxelHeadV2, ScanNetImVoxelHead,  # noqa: F401
                           SunRgbdImVoxelHead)
from .detector import ImVoxelNet, get_points                               # noqa: F401
from .boxes import (LiDARInstance3DBoxes, DepthInstance3DBoxes, limit_period, xywhr2xyxyr,  # noqa: F401
                    rotation_3d_in_axis, bbox3d2result)
from .nms import nms_gpu, nms_normal_gpu, box3d_multiclass_nms, aligned_3d_nms, boxes_iou_bev  # noqa: F401
from .evaluation import indoor_eval, average_precision, eval_det_cls, eval_map_recall  # noqa: F401
from .kitti_ap import kitti_eval, kitti_eval_coco_style, bbox2result_kitti  # noqa: F401
from .params import randomize_                                              # noqa: F401

from .data import (load_checkpoint, prepare_image, MultiViewPipeline, KittiSetOrigin, SunRgbdSetOrigin,  # noqa: F401
                   imresize_cv2_linear)
from .registry import maybe_register_into_mmdet as _reg_mmdet, register_into_mmdet  # noqa: F401

_reg_mmdet()          # opt-in (IVX_REGISTER_MMDET=1); otherwise call register_into_mmdet() explicitly

__version__ = '0.3.1'
