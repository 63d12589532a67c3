"""Drop-in for the reference's ``roi_align`` extension package (un-vendored submodule
/root/reference/.gitmodules:1-3): ``roi_align.roi_align.roi_align_3d``.

    import sys, instance_nerf_amd.roi_align as ra
    sys.modules["roi_align"] = ra; sys.modules["roi_align.roi_align"] = ra.roi_align
makes /root/reference/nerf_rcnn/model/utils.py:18 (``import roi_align``) resolve here.
"""
from . import roi_align  # noqa: F401
