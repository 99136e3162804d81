"""cpcsv_oracle — CPU fp32 restatement (the parity oracle). TEST INFRASTRUCTURE ONLY.

Pinned against the real reference: oracle/gen_golden.py imports /root/reference on CPU
(with the shims in oracle/ref_shims) and tests/test_oracle_vs_golden.py replays the
committed fixtures in tests/golden/ through this package. Every function cites the
reference file:line it restates.
"""
from .config import OracleCfg, pororo_cfg, tiny_cfg, clevr_cfg  # noqa: F401
from .nets import (StoryGenerator, CascadeStoryGenerator, FrameCritic, SegCritic,  # noqa: F401
                   StoryCritic, CondLogits, OrderCritic, init_like_reference, order_critic_state)
from .losses import (critic_loss, generator_loss, kl_term, multilabel_hit_rate, shuffle_plan, apply_shuffle)  # noqa: F401
from .ingest import image_transform, video_transform  # noqa: F401
from .step import TrainState, make_state, synthetic_batch, train_step, NoiseTape  # noqa: F401
