"""
The two board games the whole-game fixtures replay on the GPU box (no /root/reference there): the per-object
classes of ``mzx.games`` (plugin surface of games/abstract_game.py:9-105), which reproduce, observation for
observation, what the reference's games/tictactoe.py:125-351 and games/connect4.py:125-346 emit in self-play --
checked against the real files by test_reference_live.py when the reference is present.  Real game files drop in
unchanged; these exist so that tests and bench.py have the real rules without the reference tree.
"""
from mzx.games import Connect4, TicTacToe

GAMES = {"tictactoe": TicTacToe, "connect4": Connect4}
