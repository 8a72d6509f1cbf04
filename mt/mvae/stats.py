"""mt/mvae/stats.py:103-268."""
from mvae_amd.models import BatchStats, EagerBatchStatsFloat  # noqa: F401
from mvae_amd.stats import BatchStatsFloat, EpochStats, EpochStatsType  # noqa: F401
