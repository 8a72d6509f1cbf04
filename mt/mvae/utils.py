"""mt/mvae/utils.py: model-string grammar, seeds, beta schedule, CurvatureOptimizer."""
from mvae_amd.trainer import CurvatureOptimizer  # noqa: F401
from mvae_amd.utils import (canonical_name, linear_betas, parse_component_str, parse_components,  # noqa: F401
                            sampling_procedure_map, set_seeds, space_creator_map)


def setup_gpu(device):  # utils.py:51-54: cudnn flags of the reference; nothing to configure on this path
    return None
