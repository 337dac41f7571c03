"""The BASELINE workload dicts live in the package (imvoxelnet_amd/workloads.py); the tests keep their old import name."""
from imvoxelnet_amd.workloads import *  # noqa: F401,F403
