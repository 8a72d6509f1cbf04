"""Model-string grammar and small helpers of the reference (mt/mvae/utils.py), same names and error behaviour."""
from collections import defaultdict
from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch

from .components import (Component, ConstantComponent, EuclideanComponent, HyperbolicComponent, PoincareComponent,
                         SphericalComponent, StereographicallyProjectedSphereComponent, UniversalComponent)
from .sampling import (EuclideanConstantProcedure, EuclideanNormalProcedure, UniversalSamplingProcedure,
                       WrappedNormalProcedure)

# utils.py:30-48.  `c` parses and constructs as in the reference and fails in init_layers as in the reference
# (TypeError: EuclideanConstantProcedure needs a `dim` that Component.init_layers never passes).
space_creator_map = {
    "h": HyperbolicComponent,
    "u": UniversalComponent,
    "s": SphericalComponent,
    "d": StereographicallyProjectedSphereComponent,
    "p": PoincareComponent,
    "c": ConstantComponent,
    "e": EuclideanComponent,
}
sampling_procedure_map = {
    SphericalComponent: WrappedNormalProcedure,
    StereographicallyProjectedSphereComponent: WrappedNormalProcedure,
    EuclideanComponent: EuclideanNormalProcedure,
    HyperbolicComponent: WrappedNormalProcedure,
    PoincareComponent: WrappedNormalProcedure,
    ConstantComponent: EuclideanConstantProcedure,
    UniversalComponent: UniversalSamplingProcedure,
}


def set_seeds(seed: int) -> None:  # utils.py:56-59
    torch.manual_seed(seed)
    np.random.seed(seed)


def canonical_name(components: List[Component]) -> str:  # utils.py:62-75
    spaces_dims: Dict[str, Dict[int, int]] = defaultdict(lambda: defaultdict(lambda: 0))
    for component in components:
        spaces_dims[component._shortcut()[0]][component.true_dim] += 1
    parts = []
    for t in sorted(spaces_dims):
        for dim in sorted(spaces_dims[t]):
            mult = spaces_dims[t][dim]
            parts.append(f"{mult if mult > 1 else ''}{t}{dim}")
    return ",".join(parts)


def parse_component_str(space_str: str) -> Tuple[int, str, int]:  # utils.py:78-100
    s = space_str.split("-")[0]
    i = 0
    while i < len(s) and "0" <= s[i] <= "9":
        i += 1
    mult = s[:i] if i < len(s) else ""
    j = i
    while j < len(s) and "a" <= s[j] <= "z":
        j += 1
    letter = s[len(mult):j] if j < len(s) else ""
    return int(mult or "1"), letter, int(s[j:])


def parse_components(arg: str, fixed_curvature: bool) -> List[Component]:  # utils.py:103-140
    arg = arg.lower().strip()
    if not arg:
        return []
    components: List[Component] = []
    for token in (t.strip() for t in arg.split(",")):
        mult, letter, dim = parse_component_str(token)
        if mult < 1:
            raise ValueError(f"Space multiplier has to be at least 1, was: '{mult}'.")
        if dim < 1:
            raise ValueError(f"Dimension has to be at least 1, was: '{dim}'.")
        if letter not in space_creator_map:
            raise NotImplementedError(f"Unknown latent space type '{letter}'.")
        creator = space_creator_map[letter]
        for _ in range(mult):
            components.append(creator(dim, fixed_curvature, sampling_procedure=sampling_procedure_map[creator]))
    return components


def linear_betas(start: float, end: float, end_epoch: int, epochs: int) -> np.ndarray:  # utils.py:143-145
    return np.concatenate((np.linspace(start, end, num=end_epoch), end * np.ones((epochs - end_epoch,),
                                                                                 dtype=np.float32)))
