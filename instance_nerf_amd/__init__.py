"""MI355X-native instance-field NeRF render/train hot path.

The package mirrors the extension/module names of torch-ngp, the reference's
un-vendored hot-path submodule, over a C ABI
(include/inr.h) implemented with hand-written HIP kernels for gfx950.
"""
__all__ = ["raymarching", "gridencoder", "shencoder", "activation", "encoding", "ffmlp", "nerf", "scene"]


def install_aliases(include_nerf=True):
    """Registers this package's modules under the names the reference's hot-path submodule (torch-ngp,
    /root/reference/.gitmodules:4-6) and its NeRF-RCNN code import: ``raymarching``, ``gridencoder``, ``shencoder``,
    ``activation``, ``encoding``, ``ffmlp``, ``roi_align`` / ``roi_align.roi_align`` (/root/reference/nerf_rcnn/model/utils.py:18,608)
    and - with ``include_nerf`` - ``nerf``, ``nerf.network``, ``nerf.renderer``, ``nerf.utils``, ``nerf.provider``.
    After this, the reference's own ``import raymarching`` / ``from nerf.network import NeRFNetwork`` resolve to the
    HIP implementation without touching its sources.  Names that are already imported are left alone and reported."""
    import importlib
    import sys
    names = {"raymarching": ".raymarching", "gridencoder": ".gridencoder", "shencoder": ".shencoder",
             "activation": ".activation", "encoding": ".encoding", "ffmlp": ".ffmlp", "roi_align": ".roi_align",
             "roi_align.roi_align": ".roi_align.roi_align"}
    if include_nerf:
        names.update({"nerf": ".nerf", "nerf.network": ".nerf.network", "nerf.renderer": ".nerf.renderer",
                      "nerf.utils": ".nerf.utils", "nerf.provider": ".nerf.provider"})
    skipped = []
    for alias, rel in names.items():
        mod = importlib.import_module(rel, __name__)
        if alias in sys.modules and sys.modules[alias] is not mod:
            skipped.append(alias)
            continue
        sys.modules[alias] = mod
    return skipped
