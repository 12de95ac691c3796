"""``Demix.MotionCorrect`` (the reference's module name); the implementation lives in dnmf_amd."""
from dnmf_amd.Demix.MotionCorrect import MotionCorrect  # noqa: F401
