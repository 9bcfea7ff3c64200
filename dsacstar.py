"""Top-level `dsacstar` import name of the reference extension (utils/evaluation.py:11 does
`import dsacstar`); re-exports crossloc_amd.dsacstar so test_single_task-style callers are drop-in."""
from crossloc_amd.dsacstar import *  # noqa: F401,F403
from crossloc_amd.dsacstar import (RANSAC_SEED, MAX_HYPOTHESES_TRIES, MAX_REF_STEPS, backward_rgb,  # noqa: F401
                                   backward_rgb_batch, backward_rgbd, forward_rgb, forward_rgb_batch, forward_rgbd,
                                   set_image_index)
from crossloc_amd import dsacstar as _impl


def __getattr__(name):
    return getattr(_impl, name)
