"""src/NEPCore.jl:324-350"""


class NoConvergenceException(Exception):
    def __init__(self, lam=None, v=None, errmeasure=None, msg=""):
        super().__init__(msg)
        self.lam, self.v, self.errmeasure, self.msg = lam, v, errmeasure, msg


class LostOrthogonalityException(Exception):
    pass
