"""darsia_amd -- MI355X-native implementation of DarSIA's dense image hot path.

Flat namespace like ``darsia`` itself: ``darsia_amd.TVD``, ``tvd``,
``split_bregman_tvd``, ``Jacobi``, ``Solver``, ``H1_regularization``,
``backward_diff`` ... with the reference's names, arguments and error
behaviour.  All arithmetic runs in ``libdarsia_hip.so`` (hand-written HIP for
gfx950); there is no CPU fallback.
"""
from ._lib import HipError, device_count  # noqa: F401
from .device import DeviceArray, trim_device_cache  # noqa: F401
from .hostpool import trim as trim_host_pool  # noqa: F401  (recycled host result arrays: darsia_amd/hostpool.py)
from .image import ExtensiveImage, Image, OpticalImage, ScalarImage  # noqa: F401
from .geometry import (Coordinate, CoordinateArray, CoordinateSystem, Voxel, VoxelArray, full_like,  # noqa: F401
                       interpret_indexing, make_coordinate, make_voxel, ones_like, roi_to_mask,
                       to_cartesian_indexing, to_matrix_indexing, zeros_like)
from .utils.array_slice import array_slice, array_slice_argument  # noqa: F401
from .utils.dtype import convert_dtype  # noqa: F401
from .utils.derivatives import backward_diff, forward_diff, laplace  # noqa: F401
from .utils.linear_solvers.solver import Solver  # noqa: F401
from .utils.linear_solvers.jacobi import Jacobi  # noqa: F401
from .utils.linear_solvers.mg import MG  # noqa: F401
from .utils.linear_solvers.cg import CG  # noqa: F401
from .restoration.split_bregman_tvd import DeviceTVD, split_bregman_tvd  # noqa: F401
from .restoration.h1_regularization import H1_regularization  # noqa: F401
from .restoration.chambolle import denoise_tv_chambolle  # noqa: F401
from .restoration.bregman_gs import denoise_tv_bregman  # noqa: F401
from .restoration.tvd import TVD, tvd  # noqa: F401

from .color import AffineBalance, BaseBalance, apply_affine_balance  # noqa: F401
from .registration import (Patches, PiecewisePerspectiveTransform, evaluate_rbf, evaluate_rbf_on_grid,  # noqa: F401
                           get_perspective_transform)
from .imageregistration import (DiffeomorphicImageRegistration, ImageRegistration,  # noqa: F401
                                MultiscaleDiffeomorphicImageRegistration, PhaseCorrelationEstimator,
                                TranslationAnalysis)
from .concentration import (ClipModel, CombinedModel, ConcentrationAnalysis, LinearModel,  # noqa: F401
                            MonochromaticReduction, ScalingModel)

__version__ = "0.1.0"
from .corrections import (  # noqa: F401,E402
    BaseCorrection,
    CurvatureCorrection,
    DeformationCorrection,
    load_curvature_correction_config_from_dict,
    load_curvature_correction_config_from_toml,
    extract_quadrilateral_ROI,
    map_coordinates,
)
from .resizing import Resize, cv_resize, equalize_voxel_size, resize, uniform_refinement  # noqa: F401,E402
