"""Import harness for the genuine reference (runs ONLY in the build container, never on the GPU box).

The reference (/root/reference, pure Python) does not import as-is here: `cv2` is not installed and numpy >= 1.24
dropped the `np.int` / `np.float` aliases it uses.  These are ordinary Python errors (nothing was denied by the
environment).  This harness
  * registers a stand-in module object for `cv2` whose only working entry points are `fillPoly`, `line`,
    `getRotationMatrix2D` and `warpAffine` (INTER_NEAREST), all backed by the oracle's own OpenCV restatement
    (oracle/bcp_oracle.c: bco_fill_poly / bco_line / bco_rotation_matrix_2d / bco_warp_affine_nearest), and
  * restores `np.int = int`, `np.float = float`.
Everything else that then runs is the reference's own code.  Consequently golden vectors that pass through
`cv2.fillPoly` (get_pixel_footprint / pose_collides / full PlanEnv.step) pin the *rest* of the arithmetic exactly,
while the fill itself stays pinned only by the reference's known-answer tests (see bcp_oracle.h header).

Only oracle/gen_golden.py uses this module (in the build container, where /root/reference exists); the committed
fixtures under tests/golden/ are what travels.
"""
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("BCP_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "bc_gym_planning_env"))


class _Cv2Stub(types.ModuleType):
    """Module object standing in for cv2: constants resolve to 0, drawing goes through the oracle restatement."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name.isupper():
            return 0

        def _missing(*_a, **_k):
            raise NotImplementedError("cv2.%s is not available in the oracle harness" % name)
        return _missing


def _install_cv2_stub():
    import oracle as O

    cv2 = _Cv2Stub("cv2")

    def fill_poly(img, pts, color, *_a, **_k):
        value = color[0] if isinstance(color, (tuple, list)) else color
        for contour in pts:
            O.fill_poly(img, np.asarray(contour).reshape(-1, 2), int(value))
        return img

    def line(img, p0, p1, color, thickness=1, *_a, **_k):
        value = color[0] if isinstance(color, (tuple, list)) else color
        if thickness > 1:
            raise NotImplementedError("harness cv2.line only draws 1-px lines")
        O.line(img, p0, p1, int(value))
        return img

    def get_rotation_matrix_2d(center, angle, scale=1):
        return O.rotation_matrix_2d(center, angle, scale)

    def warp_affine(src, M, dsize, flags=0, borderValue=0, *_a, **_k):
        if flags != 0:
            raise NotImplementedError("harness cv2.warpAffine only does INTER_NEAREST")
        if np.asarray(src).dtype != np.uint8 or np.asarray(src).ndim != 2:
            raise NotImplementedError("harness cv2.warpAffine only takes single-channel uint8 images")
        value = borderValue[0] if isinstance(borderValue, (tuple, list)) else borderValue
        return O.warp_affine_nearest(src, np.asarray(M, dtype=np.float64), dsize, int(value))

    cv2.fillPoly = fill_poly
    cv2.line = line
    cv2.getRotationMatrix2D = get_rotation_matrix_2d
    cv2.warpAffine = warp_affine
    cv2.setNumThreads = lambda *_a: None
    cv2.getNumThreads = lambda *_a: 1
    cv2.ipp = types.SimpleNamespace(setUseIPP=lambda *_a, **_k: None)
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda *_a: None)
    sys.modules["cv2"] = cv2
    return cv2


_loaded = False


def load():
    """Make `import bc_gym_planning_env...` work; returns the package."""
    global _loaded
    if not available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    if not _loaded:
        if "cv2" not in sys.modules:
            _install_cv2_stub()
        if not hasattr(np, "int"):
            np.int = int
        if not hasattr(np, "float"):
            np.float = float
        if REFERENCE_ROOT not in sys.path:
            sys.path.insert(0, REFERENCE_ROOT)
        _loaded = True
    import bc_gym_planning_env  # noqa: F401
    return bc_gym_planning_env


class NoiseTap(object):
    """Replaces np.random.normal while active: draws z ~ N(0,1) from its own RandomState, returns loc + scale*z
    (the same arithmetic numpy's legacy normal() performs) and records every z in call order."""

    def __init__(self, seed):
        self.rng = np.random.RandomState(seed)
        self.calls = []
        self._orig = None

    def __call__(self, loc=0.0, scale=1.0, size=None):
        assert size is None
        z = float(self.rng.standard_normal())
        self.calls.append(z)
        return loc + scale * z

    def __enter__(self):
        self._orig = np.random.normal
        np.random.normal = self
        return self

    def __exit__(self, *exc):
        np.random.normal = self._orig
        return False

    def take(self):
        out, self.calls = self.calls, []
        return out
