"""
Batched-protocol versions of three reference games (tic-tac-toe, connect4, gomoku): a whole self-play shard stepped by ONE object with array
arithmetic, so that real environments stop being the host bottleneck of the batched search (SURVEY.md section 8f
row 3).  ``mzx.self_play.SelfPlay`` drives a class with ``batched = True`` through

    Game(seeds)                      one object for len(seeds) games
    reset()            -> observations [B, *observation_shape]
    step(actions [B], active=None)   -> (observations [B, ...], rewards [B], done [B]); games with active[i] false
                                        (finished earlier) are left untouched
    legal_actions()    -> int32 [B][A], each row the legal actions in increasing order, padded with -1
    to_play()          -> [B]
    reset_games(idx)   -> observations [len(idx), ...]   (optional: restarts only those games; SelfPlay.play_rounds
                                        refills finished slots through it so that every search runs at full width)

Game i of ``TicTacToeBatched`` / ``Connect4Batched`` / ``GomokuBatched`` emits, observation for observation (values AND
dtype), reward for reward, what ``Game(seed)`` of the reference's games/tictactoe.py:125-351 / games/connect4.py:125-346 /
games/gomoku.py:130-300 emits for
the same actions (tests/test_batched_games.py runs both side by side against the live files): boards as planes
[own stones of player 1, stones of player -1, side to move], a win pays 20 / 10 to the player who just moved, a game
ends on a win or a full board, players alternate 0, 1.  Only the self-play surface is provided (no rendering, no
human / expert opponents: use the reference's per-object classes for those modes).

``TicTacToe`` / ``Connect4`` are the same two games as per-object classes with the reference's plugin surface
(games/abstract_game.py:9-105): what the per-object self-play legs of bench.py and the whole-game fixtures use where
the reference tree is absent (an unmodified reference game file drops in just the same).
"""
import numpy


class _KInARow:
    """One game: generic k-in-a-row board with the plugin surface of games/abstract_game.py:9-105."""
    rows = cols = k = 0
    gravity = False
    reward_scale = 1
    int_planes = True
    end_pays = False        # gomoku: the reward is paid whenever the game ends, a full board included (gomoku.py:243)

    def __init__(self, seed=None):
        self.reset()

    def reset(self):
        self.board = numpy.zeros((self.rows, self.cols), dtype="int32")
        self.player = 1
        return self._observation()

    def to_play(self):
        return 0 if self.player == 1 else 1

    def _observation(self):
        if self.int_planes:
            mine = (self.board == 1).astype("int32")
            theirs = (self.board == -1).astype("int32")
            turn = numpy.full(self.board.shape, self.player, dtype="int32")
            return numpy.stack([mine, theirs, turn]).astype("int32")
        mine = (self.board == 1).astype("float64")
        theirs = (self.board == -1).astype("float64")
        turn = numpy.full(self.board.shape, self.player, dtype="float64")
        return numpy.stack([mine, theirs, turn])

    def legal_actions(self):
        if self.gravity:
            return [c for c in range(self.cols) if self.board[self.rows - 1, c] == 0]
        return [i for i in range(self.rows * self.cols) if self.board[i // self.cols, i % self.cols] == 0]

    def _wins(self, player):
        b = self.board == player
        R, C, k = self.rows, self.cols, self.k
        for r in range(R):
            for c in range(C):
                for dr, dc in ((0, 1), (1, 0), (1, 1), (1, -1)):
                    rr, cc = r + (k - 1) * dr, c + (k - 1) * dc
                    if 0 <= rr < R and 0 <= cc < C and all(b[r + i * dr, c + i * dc] for i in range(k)):
                        return True
        return False

    def step(self, action):
        if self.gravity:
            for r in range(self.rows):
                if self.board[r, action] == 0:
                    self.board[r, action] = self.player
                    break
        else:
            self.board[action // self.cols, action % self.cols] = self.player
        won = self._wins(self.player)
        done = won or len(self.legal_actions()) == 0
        reward = 1 if (won or (self.end_pays and done)) else 0
        self.player *= -1
        return self._observation(), reward * self.reward_scale, done

    def render(self):
        print(self.board[::-1] if self.gravity else self.board)

    def close(self):
        pass

    def action_to_string(self, action_number):
        return str(action_number)


class TicTacToe(_KInARow):
    """games/tictactoe.py:125-310, one game."""
    rows, cols, k = 3, 3, 3
    reward_scale = 20
    int_planes = True


class Connect4(_KInARow):
    """games/connect4.py:125-300, one game."""
    rows, cols, k = 6, 7, 4
    gravity = True
    reward_scale = 10
    int_planes = False


class Gomoku(_KInARow):
    """games/gomoku.py:130-300, one game: 11 x 11, five in a row, float64 planes, reward 1 when the game ends."""
    rows, cols, k = 11, 11, 5
    reward_scale = 1
    int_planes = False
    end_pays = True


PER_OBJECT = {"tictactoe": TicTacToe, "connect4": Connect4, "gomoku": Gomoku}


class _KInARowBatched:
    batched = True
    rows = cols = k = 0
    gravity = False         # connect4: a stone falls to the lowest empty cell of its column
    reward_scale = 1
    obs_dtype = "int32"
    end_pays = False        # gomoku: the reward is paid whenever the game ends (win or full board)

    def __init__(self, seeds):
        self.num_games = len(seeds)
        R, C, k = self.rows, self.cols, self.k
        lines = []
        for r in range(R):
            for c in range(C):
                for dr, dc in ((0, 1), (1, 0), (1, 1), (1, -1)):
                    rr, cc = r + (k - 1) * dr, c + (k - 1) * dc
                    if 0 <= rr < R and 0 <= cc < C:
                        lines.append([(r + i * dr) * C + (c + i * dc) for i in range(k)])
        self._lines = numpy.asarray(lines, numpy.int64)                  # [L][k] flat cell indices
        self._A = C if self.gravity else R * C
        self._arange_a = numpy.arange(self._A, dtype=numpy.int32)
        self._games = numpy.arange(self.num_games)
        self.reset()

    # ---- plugin surface -------------------------------------------------------------------------------------
    def reset(self):
        self.board = numpy.zeros((self.num_games, self.rows * self.cols), numpy.int32)
        self.player = numpy.ones(self.num_games, numpy.int32)
        self.height = numpy.zeros((self.num_games, self.cols), numpy.int64)   # stones per column (gravity games)
        return self._observation()

    def reset_games(self, games):
        """Optional refill hook of the batched protocol: restart ONLY the given games (a finished game hands its slot to
        the next one at once, like the reference actor's loop self_play.py:31-52); returns their first observations."""
        g = numpy.asarray(games, numpy.int64)
        self.board[g] = 0
        self.player[g] = 1
        self.height[g] = 0
        return self._observation()[g]

    def to_play(self):
        return numpy.where(self.player == 1, 0, 1)

    def legal_actions(self):
        if self.gravity:
            free = self.board.reshape(self.num_games, self.rows, self.cols)[:, self.rows - 1, :] == 0
        else:
            free = self.board == 0
        order = numpy.sort(numpy.where(free, self._arange_a[None, :], self._A), axis=1)
        return numpy.where(order == self._A, -1, order).astype(numpy.int32)

    def step(self, actions, active=None):
        a = numpy.asarray(actions).astype(numpy.int64)
        act = numpy.ones(self.num_games, bool) if active is None else numpy.asarray(active, bool)
        g = self._games[act]
        col = a[act]
        if self.gravity:
            row = self.height[g, col]
            full = row >= self.rows                       # the reference's loop places nothing in a full column
            row = numpy.minimum(row, self.rows - 1)
            cell = row * self.cols + col
            ok = ~full
            self.board[g[ok], cell[ok]] = self.player[g[ok]]
            self.height[g[ok], col[ok]] += 1
        else:
            self.board[g, col] = self.player[g]
        mine = self.board[:, self._lines] == self.player[:, None, None]          # [B][L][k]
        won = mine.all(2).any(1) & act
        if self.gravity:
            no_move = (self.board.reshape(self.num_games, self.rows, self.cols)[:, self.rows - 1, :] != 0).all(1)
        else:
            no_move = (self.board != 0).all(1)
        done = (won | no_move) & act
        reward = numpy.where(done if self.end_pays else won, self.reward_scale, 0).astype(numpy.int64)
        self.player = numpy.where(act, -self.player, self.player).astype(numpy.int32)
        return self._observation(), reward, done

    def close(self):
        pass

    # ---- helpers --------------------------------------------------------------------------------------------
    def _observation(self):
        b = self.board.reshape(self.num_games, self.rows, self.cols)
        out = numpy.empty((self.num_games, 3, self.rows, self.cols), self.obs_dtype)
        out[:, 0] = b == 1
        out[:, 1] = b == -1
        out[:, 2] = self.player[:, None, None]
        return out


class TicTacToeBatched(_KInARowBatched):
    """games/tictactoe.py:125-310 for a shard: 3 x 3, three in a row, int32 planes, reward 20 for a win."""
    rows, cols, k = 3, 3, 3
    reward_scale = 20
    obs_dtype = "int32"


class Connect4Batched(_KInARowBatched):
    """games/connect4.py:125-300 for a shard: 6 x 7 with gravity, four in a row, float64 planes, reward 10 for a win."""
    rows, cols, k = 6, 7, 4
    gravity = True
    reward_scale = 10
    obs_dtype = "float64"


class GomokuBatched(_KInARowBatched):
    """games/gomoku.py:130-300 for a shard: 11 x 11, five in a row, float64 planes, reward 1 when a game ends."""
    rows, cols, k = 11, 11, 5
    reward_scale = 1
    obs_dtype = "float64"
    end_pays = True


BATCHED = {"tictactoe": TicTacToeBatched, "connect4": Connect4Batched, "gomoku": GomokuBatched}


# ---- natively stepped games (include/mzx.h "Games that step NATIVELY", csrc/mzx_games.h) --------------------------------
class NativeBatchedGame:
    """
    A shard of games stepped inside the library (``mzx_game_*``): the batched plugin protocol above on a native game
    object -- and, because it carries ``native_handle``, a game ``SelfPlay.play_rounds`` plays WITHOUT returning to the
    interpreter per move (``mzx_selfplay_rounds``: search, action draw, step, history row and slot refill of every round in
    one call; ``config.native_rounds = False`` keeps the Python loop, the A/B and the parity reference of
    tests/test_native_rounds.py).  Game i equals game i of the Python class of the same name, observation for
    observation (values and dtype) and reward for reward (tests/test_native_games.py).  Subclasses fix ``kind``; the
    synthetic environment takes its geometry from ``make_native_synthetic_game``.
    """
    batched = True
    native = True
    kind = None
    observation_shape = (0, 0, 0)
    num_actions = num_players = 0
    _DTYPES = ("float32", "int32", "float64")

    def __init__(self, seeds, _backend=None):
        import ctypes

        from . import _lib

        backend = _backend or getattr(type(self), "backend", None) or _lib.default_backend()
        self.lib = backend.lib
        seeds = numpy.asarray([0 if s is None else int(s) & 0xFFFFFFFF for s in seeds], numpy.uint32)
        self.num_games = int(seeds.size)
        shape = numpy.asarray(self.observation_shape, numpy.int32)
        self.native_handle = ctypes.c_void_p()
        self.lib.check(self.lib.mzx_game_create(self.kind.encode(), self.num_games, seeds.ctypes.data, shape.ctypes.data,
                                                int(self.num_actions), int(self.num_players), ctypes.byref(self.native_handle)))
        info = (ctypes.c_int32 * 8)()
        self.lib.check(self.lib.mzx_game_info(self.native_handle, ctypes.byref(info)))
        self.shape, self.A = tuple(info[0:3]), int(info[3])
        self.obs_dtype = numpy.dtype(self._DTYPES[info[5]])
        self.reward_dtype = numpy.int64 if info[6] else numpy.float64
        self._reward = numpy.empty(self.num_games, numpy.float64)
        self._done = numpy.empty(self.num_games, numpy.uint8)

    def __del__(self):
        self.close()

    def close(self):
        handle, self.native_handle = getattr(self, "native_handle", None), None
        if handle:
            try:
                self.lib.mzx_game_destroy(handle)
            except Exception:
                pass

    def _observation(self):
        out = numpy.empty((self.num_games,) + self.shape, numpy.float32)
        self.lib.check(self.lib.mzx_game_observe(self.native_handle, out.ctypes.data))
        return out if self.obs_dtype == numpy.float32 else out.astype(self.obs_dtype)

    def reset(self):
        self.lib.check(self.lib.mzx_game_reset(self.native_handle, None, 0))
        return self._observation()

    def reset_games(self, games):
        g = numpy.ascontiguousarray(games, numpy.int32)
        self.lib.check(self.lib.mzx_game_reset(self.native_handle, g.ctypes.data, int(g.size)))
        return self._observation()[g]

    def step(self, actions, active=None):
        a = numpy.ascontiguousarray(actions, numpy.int64)
        act = None if active is None else numpy.ascontiguousarray(active, numpy.uint8)
        self.lib.check(self.lib.mzx_game_step(self.native_handle, a.ctypes.data, None if act is None else act.ctypes.data,
                                              self._reward.ctypes.data, self._done.ctypes.data))
        return self._observation(), self._reward.astype(self.reward_dtype), self._done.astype(bool)

    def legal_actions(self):
        out = numpy.empty((self.num_games, self.A), numpy.int32)
        self.lib.check(self.lib.mzx_game_legal_actions(self.native_handle, out.ctypes.data))
        return out

    def to_play(self):
        out = numpy.empty(self.num_games, numpy.int32)
        self.lib.check(self.lib.mzx_game_to_play(self.native_handle, out.ctypes.data))
        return out.astype(numpy.int64)


class TicTacToeNative(NativeBatchedGame):
    """games/tictactoe.py:125-310 for a shard, stepped natively."""
    kind = "tictactoe"


class Connect4Native(NativeBatchedGame):
    """games/connect4.py:125-300 for a shard, stepped natively."""
    kind = "connect4"


class GomokuNative(NativeBatchedGame):
    """games/gomoku.py:130-300 for a shard, stepped natively."""
    kind = "gomoku"


def make_native_synthetic_game(observation_shape, num_actions, num_players=1):
    """``mzx.synthetic.make_synthetic_batched_game`` stepped natively: the fixed-shape environment of the metric."""
    class SyntheticNative(NativeBatchedGame):
        kind = "synthetic"

    SyntheticNative.observation_shape = tuple(int(x) for x in observation_shape)
    SyntheticNative.num_actions, SyntheticNative.num_players = int(num_actions), int(num_players)
    return SyntheticNative


NATIVE = {"tictactoe": TicTacToeNative, "connect4": Connect4Native, "gomoku": GomokuNative}
