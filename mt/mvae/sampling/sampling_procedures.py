"""mt/mvae/sampling/sampling_procedures.py:31-206."""
from mvae_amd.sampling import (EuclideanConstantProcedure, EuclideanNormalProcedure, SamplingProcedure,  # noqa: F401
                               UniversalSamplingProcedure, WrappedNormalProcedure)
