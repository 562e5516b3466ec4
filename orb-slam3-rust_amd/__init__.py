"""orb-slam3-rust_amd — MI355X-native (gfx950) hot path of jurmy24/orb-slam3-rust:
stereo ORB extract + match + triangulate and visual local bundle adjustment, as hand-written HIP
kernels behind the C ABI of include/orbx.h.  See DESIGN.md / INTEGRATION.md.

Import as `orb_slam3_rust_amd` (the shim at the repo root maps the hyphenated directory name).
"""
from . import dist, synth  # noqa: F401
from .api import (  # noqa: F401
    ABI_SYMBOLS, BA_OBS, BA_OBS32, ba_obs_to_obs32, DMATCH, KEYPOINT, NN_RATIO, TH_HIGH, TH_LOW, CameraModel, FeatureSet, GlobalBAConfig, Handle,
    LocalBAConfigLM, OrbxError, StereoFrame, StereoProcessor, VisualBAProblemData, VisualBAResultData,
    VisualObservation, bf_match_crosscheck, descriptor_distance, flatten_ba_problem, load_library,
    solve_visual_ba, GlobalBAObservation, GlobalBAProblemData, GlobalBAResult, flatten_global_ba_problem, se3_inverse,
    solve_global_ba, OrbVocabulary, EurocDataset, png_decode_gray8, LocalInertialBAConfig, InertialVisualObs, ImuEdgeData, InertialBAProblemData,
    InertialBAResultData, flatten_inertial_ba_problem, solve_inertial_ba, MapSnapshot, local_bundle_adjustment, run_global_ba, KeyFrame, BaBatch)
from .build import LIB_PATH, build  # noqa: F401
