"""eam_rl4co_amd: MI355X-native construction rollout (TSP / CVRP and the sibling routing envs CVRPTW, SDVRP, PCTSP, OP
+ AttentionModel decode) behind the RL4CO interfaces of Tarseus/eam-rl4co.  See DESIGN.md."""
from .envs import (CVRPEnv, CVRPGenerator, CVRPTWEnv, CVRPTWGenerator, OPEnv, OPGenerator, PCTSPEnv, PCTSPGenerator, RL4COEnvBase, SDVRPEnv, SPCTSPEnv,  # noqa: F401
                   TSPEnv, TSPGenerator, get_env)  # noqa: F401
from .policy import (AttentionModelDecoder, AttentionModelEncoder, AttentionModelPolicy, GraphedRollout, SymNCOPolicy,  # noqa: F401
                     load_reference_checkpoint, random_policy, rollout)
from .attention import PointerAttention, scaled_dot_product_attention  # noqa: F401
from .evolution import EA, EACvrpDraws, EAPrizeDraws, EADraws, evolution_worker, generate_batch_population  # noqa: F401
from .tensordict_lite import TensorDict  # noqa: F401
from .utils import batchify, gather_by_index, unbatchify  # noqa: F401

__version__ = "0.1.0"
