"""
TEST DOUBLES of two reference plugin games, written from scratch so the GPU box
(no /root/reference) can replay the whole-game fixtures: a generic k-in-a-row
board game exposing the plugin surface of games/abstract_game.py:9-105.

They reproduce, observation for observation, what the reference's
games/tictactoe.py:125-351 and games/connect4.py:125-346 emit in self-play
(checked against the real files by test_reference_live.py when the reference is
present).  They are NOT part of the product: real game files drop in unchanged.
"""
import numpy


class _KInARow:
    rows = cols = k = 0
    gravity = False
    reward_scale = 1
    int_planes = True

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
        reward = 1 if won else 0
        self.player *= -1
        return self._observation(), reward * self.reward_scale, done

    def render(self):
        print(self.board[::-1] if self.gravity else self.board)

    def close(self):
        pass

    def action_to_string(self, action_number):
        return str(action_number)


class TicTacToe(_KInARow):
    rows, cols, k = 3, 3, 3
    reward_scale = 20
    int_planes = True


class Connect4(_KInARow):
    rows, cols, k = 6, 7, 4
    gravity = True
    reward_scale = 10
    int_planes = False


GAMES = {"tictactoe": TicTacToe, "connect4": Connect4}
