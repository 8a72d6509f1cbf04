"""Step / epoch statistics (mt/mvae/stats.py:103-268) without the per-step device->host syncs.

The fused step accumulates batch sums of bce, kl, elbo and per-component kl on the device (k_dec1_bwd).
`BatchStatsFloat` keeps the reference's field names but reads the device record lazily, once, when a field is first
accessed; `EpochStats` is built from the running sums the host reads once per epoch.
"""
from typing import Dict, List, Optional

EpochStatsType = Dict[str, float]


def _to_print(stats) -> EpochStatsType:  # stats.py:103-112
    return {
        "bce": stats.bce, "kl": stats.kl, "elbo": stats.elbo,
        "ll": 0.0 if stats.log_likelihood is None else stats.log_likelihood,
        "mi": 0.0 if stats.mutual_info is None else stats.mutual_info,
        "cov_norm": 0.0 if stats.cov_norm is None else stats.cov_norm,
        "beta": stats.beta,
    }


class BatchStatsFloat:
    """Lazy view of the LAST step's record (valid until the next step overwrites it; .item()-free until read)."""

    def __init__(self, engine, beta: float, log_likelihood=None, mutual_info=None, cov_norm=None) -> None:
        self._engine = engine
        self._rec = None
        self.beta = beta
        self._ll, self._mi, self._cn = log_likelihood, mutual_info, cov_norm

    def _get(self):
        if self._rec is None:
            self._rec = self._engine.read_stats()["last"]
        return self._rec

    @property
    def bce(self) -> float:
        return self._get()["bce"]

    @property
    def kl(self) -> float:
        return self._get()["kl"]

    @property
    def elbo(self) -> float:
        return self._get()["elbo"]

    @property
    def component_kl(self) -> List[float]:
        return self._get()["component_kl"]

    @property
    def log_likelihood(self) -> Optional[float]:
        return None if self._ll is None else float(self._ll)

    @property
    def mutual_info(self) -> Optional[float]:
        return None if self._mi is None else float(self._mi)

    @property
    def cov_norm(self) -> Optional[float]:
        return None if self._cn is None else float(self._cn)

    def to_print(self) -> EpochStatsType:
        return _to_print(self)


class EpochStats:
    """stats.py:215-268: sums over the epoch divided by the dataset length."""

    def __init__(self, sums: Dict[str, object], length: int, beta: float, log_likelihood: float = 0.0,
                 mutual_info: float = 0.0, cov_norm: float = 0.0) -> None:
        self.bce = sums["bce"] / length
        self.kl = sums["kl"] / length
        self.elbo = sums["elbo"] / length
        self.component_kl = [k / length for k in sums["component_kl"]]
        self.log_likelihood = log_likelihood / length
        self.mutual_info = mutual_info / length
        self.cov_norm = cov_norm / length
        self.beta = beta

    def to_print(self) -> EpochStatsType:
        return _to_print(self)
