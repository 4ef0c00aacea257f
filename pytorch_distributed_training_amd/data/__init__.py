from .synthetic import SyntheticPairDataset  # noqa: F401
from .collate import PadCollator  # noqa: F401
from .samplers import DistributedShardSampler, ShardedBatchSampler  # noqa: F401
from .loader import DeviceLoader, shard_dataloader  # noqa: F401
