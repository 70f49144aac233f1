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

    # rocket_fft (pocketfft bindings for numba) is only needed by candidate selection; its
    # functions are reached through numba overloads, which do not exist here (see
    # install_selection_glue)
    rf = types.ModuleType("rocket_fft")
    rf.pocketfft = types.SimpleNamespace()
    rfo = types.ModuleType("rocket_fft.overloads")
    for n in "decrease_shape get_fct increase_shape ndshape_and_axes resize zeropad_or_crop".split():
        setattr(rfo, n, None)
    rf.overloads = rfo
    sys.modules.update({"rocket_fft": rf, "rocket_fft.overloads": rfo})

    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)

    # FragmentContainer.slice is a numba overload (scoring/utils.py:413-475); bind its impl.
    from alphadia.search.jitclasses.fragment_container import FragmentContainer
    from alphadia.search.scoring import utils as su

    FragmentContainer.slice = su.slice(FragmentContainer, None)


def install_selection_glue() -> None:
    """Bind plain-NumPy stand-ins for the three numba overloads candidate selection calls.

    In the reference these are ``@overload`` implementations compiled by Numba
    (selection/fft.py:119-212, selection/utils.py:24-46); without Numba the module-level names
    raise ``NumbaContextOnly``.  The stand-ins do what the overload bodies do with NumPy:
    ``rfft2`` / ``irfft2`` from ``np.fft`` replace pocketfft (the reference's own test equates the
    two at 1e-3, tests/unit_tests/search/selection/test_fft.py), results cast to float32.
    """
    install()
    from alphadia.search.selection import fft, selection

    def convolve_fourier(dense, kernel):
        k0, k1 = kernel.shape
        delta0, delta1 = -k0 // 2, -k1 // 2
        out = np.zeros_like(dense)
        shape = dense.shape[-2:]
        fourier_filter = np.fft.rfft2(kernel.astype(np.float32), shape).astype(np.complex64)
        flat_in = dense.reshape((-1,) + shape)
        flat_out = out.reshape((-1,) + shape)
        for i in range(flat_in.shape[0]):
            spec = np.fft.rfft2(flat_in[i]).astype(np.complex64)
            layer = np.fft.irfft2(spec * fourier_filter, shape).astype(np.float32)
            o = flat_out[i]
            o[delta0:, delta1:] = layer[:-delta0, :-delta1]
            o[:delta0, delta1:] = layer[-delta0:, :-delta1]
            o[delta0:, :delta1] = layer[:-delta0, -delta1:]
            o[:delta0, :delta1] = layer[-delta0:, -delta1:]
        return out

    def assemble_isotope_mz(mono_mz, charge, isotope_intensity):
        offset = np.arange(len(isotope_intensity)) * 1.0033548350700006 / charge
        isotope_mz = np.zeros(len(isotope_intensity), dtype=np.float32)
        isotope_mz[:] = mono_mz
        isotope_mz += offset
        return isotope_mz

    fft.convolve_fourier = convolve_fourier
    selection.assemble_isotope_mz = assemble_isotope_mz
