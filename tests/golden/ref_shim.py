"""Import shim that lets the *reference* hot path run as plain Python.

TEST INFRASTRUCTURE - used only by ``tests/golden/make_golden.py`` in the build
container, where ``/root/reference`` is mounted read-only.  It never travels to
the GPU box and nothing in the product imports it.

It stubs THIRD-PARTY modules that are not installed here (``numba``,
``alphatims``, ``alpharaw``); no reference source is copied.  With the stubs in
place ``alphadia.search.jitclasses``, ``alphadia.search.scoring`` and
``alphadia.fragcomp`` import unchanged from ``/root/reference`` and execute
under NumPy.

Fidelity caveats of goldens produced this way:

1. NumPy 2 weak-scalar promotion keeps ``float32 (op) python_float`` in float32
   where Numba types the literal as float64.
2. ``np.sum`` / ``np.mean`` are pairwise in NumPy and sequential in Numba.
3. ``argsort`` tie order is implementation defined.
4. The reference pins ``numpy<2``; this container has 2.2.
"""

from __future__ import annotations

import contextlib
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


def _ident(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


class _T:
    """Stands in for numba type objects (nb.float32, nb.types.X, nb.float32[:, ::1] ...)."""

    def __init__(self, name, dt=None):
        self.name = name
        if dt is not None:
            self.dtype = np.dtype(dt)

    def __getitem__(self, i):
        return self

    def __call__(self, *a, **k):
        if len(a) == 1 and isinstance(a[0], (int, float, np.generic)):
            return a[0]
        return self

    def __getattr__(self, k):
        if k.startswith("__") or k == "dtype":
            raise AttributeError(k)
        return _T(self.name + "." + k)


def install(reference_root: str = REFERENCE_ROOT) -> None:
    if "numba" in sys.modules and getattr(sys.modules["numba"], "_adh_stub", False):
        return
    nb = types.ModuleType("numba")
    nb._adh_stub = True
    nb.njit = nb.jit = _ident
    nb.TypingError = TypeError
    for n in (
        "uint8 uint16 uint32 uint64 int8 int16 int32 int64 float32 float64 complex64".split()
    ):
        setattr(nb, n, _T(n, n))
    nb.boolean = nb.bool_ = _T("bool", "bool")
    nb.types = _T("types")
    nb.core = _T("core")

    def jitclass(spec=None):
        def wrap(cls):
            cls.class_type = types.SimpleNamespace(instance_type=cls)
            return cls

        return wrap(spec) if isinstance(spec, type) else wrap

    nb.experimental = types.ModuleType("numba.experimental")
    nb.experimental.jitclass = jitclass
    nb.typed = types.SimpleNamespace(
        List=types.SimpleNamespace(empty_list=lambda t: []),
        Dict=types.SimpleNamespace(empty=lambda **k: {}),
    )
    nb.objmode = contextlib.nullcontext()
    ext = types.ModuleType("numba.extending")
    ext.overload = ext.overload_method = lambda *a, **k: (lambda f: f)
    nb.extending = ext
    core = types.ModuleType("numba.core")
    core.types = _T("types")
    sys.modules.update(
        {
            "numba": nb,
            "numba.experimental": nb.experimental,
            "numba.extending": ext,
            "numba.core": core,
        }
    )

    at = types.ModuleType("alphatims")
    atu = types.ModuleType("alphatims.utils")
    atb = types.ModuleType("alphatims.bruker")

    def pjit(*a, **k):
        def deco(f):
            def run(it, *args):
                for i in it:
                    f(i, *args)

            run.py_func = f
            return run

        if len(a) == 1 and callable(a[0]):
            return deco(a[0])
        return deco

    atu.pjit = pjit
    atu.set_threads = lambda n: n
    atb.TimsTOF = type("TimsTOF", (), {})
    at.utils = atu
    at.bruker = atb
    at.__version__ = "stub"
    sys.modules.update({"alphatims": at, "alphatims.utils": atu, "alphatims.bruker": atb})

    for m, names in {
        "alpharaw.ms_data_base": ["MSData_Base"],
        "alpharaw.mzml": ["MzMLReader"],
        "alpharaw.sciex": ["SciexWiffData"],
        "alpharaw.thermo": ["ThermoRawData"],
    }.items():
        mod = types.ModuleType(m)
        for n in names:
            setattr(mod, n, type(n, (), {"__init__": lambda self, *a, **k: None}))
        sys.modules[m] = mod
    sys.modules["alpharaw"] = types.ModuleType("alpharaw")

    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)

    # FragmentContainer.slice is a numba overload (scoring/utils.py:413-475); bind its impl.
    from alphadia.search.jitclasses.fragment_container import FragmentContainer
    from alphadia.search.scoring import utils as su

    FragmentContainer.slice = su.slice(FragmentContainer, None)
