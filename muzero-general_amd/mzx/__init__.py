"""
mzx -- MI355X-native batched MuZero self-play engine (hot path only).

Host-side mirror of the reference's operator surface for the self-play path
(/root/reference/self_play.py, models.py): ``mzx.models.MuZeroNetwork``,
``mzx.self_play.{MCTS, SelfPlay, GameHistory}``.  All compute goes through the
C-ABI shared library built from ``muzero-general_amd/csrc`` (``include/mzx.h``);
there is no CPU fallback -- calling a compute entry point without the HIP
library or without a GPU raises.
"""
__version__ = "0.1.0"
