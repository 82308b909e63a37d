"""Overlay that puts the MI355X path under an UNMODIFIED checkout of the reference.

The reference binds its hot path by module attribute: `from renderer import rendering`,
`from networks import Embedding, MVSNeRF, MVSNet, MVSNeRF_G, DyMVSNeRF_G, <discriminators>`,
`from utils import build_rays, visualize_depth, projection_from_ndc`, `from losses import ...`
(/root/reference/train.py:36-44; networks.py:25-26; renderer.py:20).  This package ships no module
named `networks`, `utils`, `renderer` or `losses` - it shadows nothing.  `install()` imports the
CALLER'S OWN modules of those names and rebinds, inside them, only the names of the rendering
path to the HIP implementations (zest_renderer, zest_networks, zest_utils, zest_losses); every
other name - discriminators, visualisation helpers, image-space losses, data loaders - stays the
caller's.  A script that runs afterwards (`from networks import ...`) picks the rebound names up.

    import zest_dropin; zest_dropin.install()           # then: import train
    python -m zest_dropin train.py --config ...          # same, through runpy

`inplace_abn` (a CUDA-only extension the reference's networks.py imports at line 23) does not
exist on ROCm: unless it is importable, install() registers a stand-in module whose InPlaceABN is
zest_networks.ActivatedBatchNorm (batch norm + leaky ReLU 0.01, same parameter names), so the
caller's networks.py imports and its checkpoints load.
"""
import importlib
import os
import runpy
import sys
import types

__all__ = ["install", "uninstall", "PATH_NAMES", "main"]

# caller module -> (zest module, names rebound in the caller's module)
PATH_NAMES = {
    "utils": ("zest_utils", ("build_rays", "build_rays_dy", "build_rays_base", "get_rays_mvs", "get_ndc_coordinate",
                             "index_point_feature", "build_color_volume", "homo_warp", "projection_from_ndc")),
    "renderer": ("zest_renderer", ("rendering", "raw2outputs", "raw2outputs_blending", "raw2alpha", "depth2dist",
                                   "compute_2d_prob")),
    "networks": ("zest_networks", ("Embedding", "Renderer", "Renderer_linear", "MVSNeRF", "ConvBnReLU", "ConvBnReLU3D",
                                   "FeatureNet", "CostRegNet", "MVSNet", "MVSNeRF_G", "DyMVSNeRF_G")),
    "losses": ("zest_losses", ("distortion_loss",)),
}
# names a caller module pulled in with `from X import name` before the overlay ran: rebind the copies too
_REIMPORTED = {
    "networks": (("zest_utils", ("homo_warp", "build_rays", "build_rays_dy")), ("zest_renderer", ("rendering",))),
    "renderer": (("zest_utils", ("index_point_feature", "build_color_volume")),),
}
_saved = []          # (module, name, had, old) for uninstall()


def _stub_inplace_abn():
    try:
        importlib.import_module("inplace_abn")
        return False
    except ImportError:
        import zest_networks
        m = types.ModuleType("inplace_abn")
        m.__doc__ = "zest_dropin stand-in: InPlaceABN -> zest_networks.ActivatedBatchNorm (ROCm has no inplace_abn)"
        m.InPlaceABN = zest_networks.ActivatedBatchNorm
        m.ABN = zest_networks.ActivatedBatchNorm
        sys.modules["inplace_abn"] = m
        return True


def _bind(mod, name, value):
    _saved.append((mod, name, hasattr(mod, name), getattr(mod, name, None)))
    setattr(mod, name, value)


def install(reference_dir=None, modules=("utils", "renderer", "networks", "losses"), stub_inplace_abn=True):
    """Import the caller's `modules` (from `reference_dir` if given, else from sys.path as it stands)
    and rebind the rendering path's names in them.  Returns {module name: [rebound names]}.
    Raises ImportError if one of the caller's modules cannot be imported, and RuntimeError if a
    module found under one of those names is this package's own (nothing to overlay)."""
    here = os.path.dirname(os.path.abspath(__file__))
    if reference_dir is not None:
        reference_dir = os.path.abspath(reference_dir)
        if reference_dir not in sys.path:
            sys.path.insert(0, reference_dir)
    if stub_inplace_abn:
        _stub_inplace_abn()
    done, targets = {}, {}
    for name in modules:                                             # first import all of the caller's modules ...
        target = importlib.import_module(name)
        if os.path.dirname(os.path.abspath(getattr(target, "__file__", "") or "")) == here:
            raise RuntimeError("zest_dropin: module %r resolves to this package; put the reference checkout "
                               "on sys.path (or pass reference_dir)" % name)
        targets[name] = target
    for name in modules:                                             # ... then rebind (uninstall() restores their own names)
        zest_name, names = PATH_NAMES[name]
        target = targets[name]
        zest = importlib.import_module(zest_name)
        for n in names:
            _bind(target, n, getattr(zest, n))
        for zn, copies in _REIMPORTED.get(name, ()):
            src = importlib.import_module(zn)
            for n in copies:
                if hasattr(target, n):
                    _bind(target, n, getattr(src, n))
        target.__zest_dropin__ = sorted(names)
        done[name] = sorted(names)
    return done


def uninstall():
    """Undo install(): the caller's modules get their own names back."""
    while _saved:
        mod, name, had, old = _saved.pop()
        if had:
            setattr(mod, name, old)
        elif hasattr(mod, name):
            delattr(mod, name)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print("usage: python -m zest_dropin SCRIPT.py [script arguments]\n"
              "runs SCRIPT (e.g. the reference's train.py / test.py) with the MI355X rendering path bound into its "
              "own networks / utils / renderer / losses modules")
        return 0 if argv else 2
    script = os.path.abspath(argv[0])
    install(reference_dir=os.path.dirname(script))
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
