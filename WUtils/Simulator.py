from dnmf_amd.WUtils.Simulator import *  # noqa: F401,F403
from dnmf_amd.WUtils.Simulator import generate_video, generate_gp_motion, simulate_exponential_traces  # noqa: F401
