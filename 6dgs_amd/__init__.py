"""6dgs_amd -- MI355X-native 6DGS pose-estimation inference path.

Drop-in for the reference's hot path (mbortolon97/6dgs): the three callables the reference driver
uses keep their names and signatures,

    generate_all_possible_rays     (pose_estimation/sampling.py:127)
    IdentificationModule           (pose_estimation/identification_module.py:10)
    test_pose_estimation           (pose_estimation/test.py:23)

while ray emission, the ray MLP / key cache, the ray<->image attention scorer, top-k selection and
the line-intersection pose solve run as hand-written HIP kernels for gfx950 behind the C ABI of
include/sixdgs.h.  PyTorch-ROCm owns device memory and streams and runs the two frozen image-side
networks (DINOv2 backbone, camera-up CNN).

The directory name starts with a digit, so import it with importlib or through the alias package:

    import importlib; sixdgs = importlib.import_module("6dgs_amd")
    import sixdgs_amd                      # same module object
"""
from .scene import GaussianModel, GaussianScene  # noqa: F401
from .sampling import generate_all_possible_rays  # noqa: F401
from .identification_module import IdentificationModule  # noqa: F401
from .test import test_pose_estimation  # noqa: F401
from .scene import CameraInfo  # noqa: F401
from .distance_based_loss import DistanceBasedScoreLoss  # noqa: F401
from .train import train_id_module  # noqa: F401

__all__ = ["GaussianModel", "GaussianScene", "CameraInfo", "generate_all_possible_rays", "IdentificationModule",
           "test_pose_estimation", "DistanceBasedScoreLoss", "train_id_module"]
