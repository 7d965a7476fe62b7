"""colbert_amd -- MI355X-native MaxSim rerank path (drop-in for wuyaoxuehun/colbert's
``BaseModel.score`` + ``ColbertRanker.rank_forward``).  Importing this package loads libmaxsim.so and fails
loudly if it has not been built."""
from . import _lib
from .ranker import ColbertRanker
from .retriever import retrieve_batch
from .scoring import MaxSimModel, score
from .sharded import ShardedRanker, load_shard

__all__ = ["ColbertRanker", "MaxSimModel", "score", "retrieve_batch", "ShardedRanker", "load_shard", "_lib"]
