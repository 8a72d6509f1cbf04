"""mt/mvae/sampling/__init__.py."""
from .sampling_procedures import (EuclideanConstantProcedure, EuclideanNormalProcedure, SamplingProcedure,
                                  UniversalSamplingProcedure, WrappedNormalProcedure)

__all__ = ["SamplingProcedure", "EuclideanConstantProcedure", "EuclideanNormalProcedure", "WrappedNormalProcedure",
           "UniversalSamplingProcedure"]
