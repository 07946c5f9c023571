"""The C-ABI library loads on a CPU-only box and exports every symbol include/pisces_hip.h declares
(no compute calls here: those need a GPU and live in the -m gpu tests)."""
import ctypes as C
import os
import re

import numpy as np

from pisces_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "pisces_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pisces_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pisces_amd import _native
    declared = _declared_functions()
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(_native.lib, name), f"{name} declared in include/pisces_hip.h but not exported"
    assert sorted(_native.EXPORTS) == declared
    assert _native.lib.pisces_hip_abi_version() == _abi.ABI_VERSION


def _split_args(arglist):
    arglist = arglist.strip()
    return [] if arglist in ("", "void") else [a.strip() for a in arglist.split(",")]


def test_managed_bindings_cover_the_header():
    """dotnet/NativeMethods.cs (the [DllImport]s a maintainer of the reference adds) cannot be compiled in this image: keep it
    honest against the header at least -- every declared entry bound, with as many arguments as the C declaration has."""
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "pisces_hip.h")).read(), flags=re.S)
    c_args = {m.group(1): len(_split_args(m.group(2))) for m in re.finditer(r"\b(pisces_hip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S)}
    assert sorted(c_args) == _declared_functions()
    cs = open(os.path.join(ROOT, "dotnet", "NativeMethods.cs")).read()
    cs = re.sub(r"//[^\n]*", "", cs)
    bound = {}
    for m in re.finditer(r"\[DllImport\(Lib[^\]]*\)\]\s*public static extern \w+ (pisces_hip_[a-z0-9_]+)\(([^;]*)\);", cs):
        args = re.sub(r"\[[^\]]*\]", "", m.group(2))   # [Out], [MarshalAs(...)]
        bound.setdefault(m.group(1), set()).add(len(_split_args(args)))
    missing = sorted(set(c_args) - set(bound))
    assert not missing, f"declared in include/pisces_hip.h, no [DllImport] in dotnet/NativeMethods.cs: {missing}"
    assert not set(bound) - set(c_args), f"[DllImport]s of entries the header does not declare: {sorted(set(bound) - set(c_args))}"
    for name, counts in bound.items():   # (overloads bind one entry twice: byte[] and IntPtr for the file bytes)
        assert counts == {c_args[name]}, f"{name}: {c_args[name]} arguments in the header, {sorted(counts)} in NativeMethods.cs"


_C_SIZES = {"int32_t": 4, "uint32_t": 4, "float": 4, "double": 8, "int16_t": 2, "uint16_t": 2, "uint8_t": 1, "int8_t": 1, "int64_t": 8, "uint64_t": 8}
_CS_SIZES = {"int": 4, "uint": 4, "float": 4, "double": 8, "short": 2, "ushort": 2, "byte": 1, "sbyte": 1, "long": 8, "ulong": 8, "IntPtr": 8}


def _c_struct_fields(text):
    """{struct name: [size of every primitive field in declaration order]} of the header (arrays flattened, pointers 8 bytes)."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"typedef struct (\w+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        sizes = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            dm = re.match(r"(?:const\s+)?(\w+)\s*(\*?)\s*(.+)$", decl, flags=re.S)
            assert dm, decl
            ctype, star, names = dm.group(1), dm.group(2), dm.group(3)
            for name in names.split(","):
                name = name.strip()
                ptr = bool(star) or name.startswith("*")
                am = re.search(r"\[(\d+)\]", name)
                n = int(am.group(1)) if am else 1
                if ptr:
                    sizes += [8] * n
                elif ctype in out:          # a struct by value
                    sizes += out[ctype] * n
                else:
                    sizes += [_C_SIZES[ctype]] * n
        out[m.group(1)] = sizes
    return out


def _cs_struct_fields(text):
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for m in re.finditer(r"public (?:unsafe )?struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        sizes = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            dm = re.match(r"public\s+(fixed\s+)?(\w+)\s*(\*?)\s*(.+)$", decl, flags=re.S)
            assert dm, decl
            fixed, cstype, star, names = dm.group(1), dm.group(2), dm.group(3), dm.group(4)
            for name in names.split(","):
                name = name.strip()
                if fixed:
                    sizes += [_CS_SIZES[cstype]] * int(re.search(r"\[(\d+)\]", name).group(1))
                elif star:
                    sizes.append(8)
                else:
                    sizes.append(_CS_SIZES[cstype])
        out[m.group(1)] = sizes
    return out


def test_managed_structs_have_the_headers_fields_in_order():
    """The [StructLayout(Sequential)] structs of dotnet/NativeMethods.cs against include/pisces_hip.h, field by field: the same number of
    primitive fields, of the same widths, in the same order (a managed struct that lost a field, or has two swapped, marshals garbage
    without any error — nothing compiles the C# here)."""
    c = _c_struct_fields(open(os.path.join(ROOT, "include", "pisces_hip.h")).read())
    cs = _cs_struct_fields(open(os.path.join(ROOT, "dotnet", "NativeMethods.cs")).read())
    must = {"PiscesHipConfig", "PiscesCalledAllele", "PiscesReadBatch", "PiscesCandidate", "PiscesVcfConfig", "PiscesVcfPadState", "PiscesBgzfBlock",
            "PiscesTile", "PiscesTileResult"}
    assert must <= set(c) and must <= set(cs), (sorted(must - set(c)), sorted(must - set(cs)))
    for name in sorted(set(c) & set(cs)):
        assert cs[name] == c[name], f"{name}: header field widths {c[name]}, NativeMethods.cs {cs[name]}"
    # and the header against the ctypes mirror the tests drive the library with
    assert sum(c["PiscesHipConfig"]) == C.sizeof(_abi.PiscesHipConfig) and sum(c["PiscesCandidate"]) + 2 * 0 <= C.sizeof(_abi.PiscesCandidate)


def test_struct_layouts_match_header():
    assert _abi.CALLED_ALLELE_DTYPE.itemsize == 64
    assert _abi.TILE_DTYPE.itemsize == 24 and _abi.TILE_RESULT_DTYPE.itemsize == 48
    assert C.sizeof(_abi.PiscesHipConfig) == 4 * 41
    assert C.sizeof(_abi.PiscesCandidate) == 56
    assert C.sizeof(_abi.PiscesBgzfBlock) == 32
    d = _abi.CALLED_ALLELE_DTYPE
    assert d.fields["strand_bias_score"][1] == 48 and d.fields["genotype_qscore"][1] == 56
    assert d.fields["filter_bits"][1] == 60 and d.fields["info"][1] == 62


def test_default_config_matches_library_and_reference_defaults():
    from pisces_amd import _native
    c = _abi.PiscesHipConfig()
    assert _native.lib.pisces_hip_default_config(C.byref(c)) == 0
    py = _abi.default_config()
    for name, _ in _abi.PiscesHipConfig._fields_:
        if name == "reserved":
            continue
        a, b = getattr(c, name), getattr(py, name)
        assert (list(a) == list(b)) if hasattr(a, "__len__") else (a == b), name
    # src/lib/Pisces.Domain/Options/VariantCallingParameters.cs:57-107
    assert (c.min_base_call_quality, c.max_variant_qscore, c.min_variant_qscore, c.variant_qscore_filter) == (20, 100, 20, 30)
    assert (c.min_coverage, c.block_size, c.strand_bias_model) == (10, 1000, _abi.SB_EXTENDED)
    assert np.float32(c.min_frequency) == np.float32(0.01) and np.float32(c.rmxn_frequency_limit) == np.float32(0.35)


def test_create_without_gpu_fails_loudly_not_silently():
    """On a box without a HIP device create() must return an error (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        return
    from pisces_amd import _native
    h = C.c_void_p()
    cfg = _abi.default_config()
    rc = _native.lib.pisces_hip_create(C.byref(cfg), 0, C.byref(h))
    assert rc == _abi.E_DEVICE and not h.value
    assert b"HIP device" in _native.lib.pisces_hip_last_error(None)


def test_graft_entry_build_succeeds():
    """The driver's build check: __graft_entry__.build() compiles (or finds up to date) the HIP library and the oracle and imports the package."""
    import __graft_entry__ as g
    g.build()


def _comm_library_in_a_fresh_process(prelude, env=None):
    """pisces_hip_comm_library binds RCCL once per process: every case gets a process of its own."""
    import subprocess
    import sys
    code = prelude + "\nfrom pisces_amd import engine\ntry:\n    print('OK ' + engine.HipVariantCaller.comm_library())\nexcept Exception as e:\n    print('ERR ' + str(e))\n"
    e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    e.pop("PISCES_HIP_RCCL_PATH", None)
    e.update(env or {})
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith(("OK ", "ERR "))]
    assert lines, (out.stdout[-500:], out.stderr[-500:])
    return lines[-1]


def test_rccl_lookup_order():
    """surface_comm.inc.h binds RCCL at run time, in this order: PISCES_HIP_RCCL_PATH (and nothing else when it is set), a librccl.so the
    process has mapped already (a PyTorch host's own copy: one RCCL per process), the loader's path / /opt/rocm/lib.  No GPU needed:
    binding is dlopen + dlsym."""
    import importlib.util
    system = "/opt/rocm/lib/librccl.so"
    if not os.path.exists(system):
        import pytest
        pytest.skip("no system RCCL in this image")
    # 1. the environment decides, also when what it names cannot be loaded (no fall-through to another copy)
    r = _comm_library_in_a_fresh_process("", {"PISCES_HIP_RCCL_PATH": "/nonexistent/librccl.so"})
    assert r.startswith("ERR ") and "PISCES_HIP_RCCL_PATH=/nonexistent/librccl.so" in r, r
    r = _comm_library_in_a_fresh_process("", {"PISCES_HIP_RCCL_PATH": system})
    assert r == "OK PISCES_HIP_RCCL_PATH: " + system, r
    # 2. a copy that is mapped already wins over the default names — and the environment wins over it
    spec = importlib.util.find_spec("torch")
    torch_rccl = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so") if spec and spec.origin else None
    if torch_rccl and os.path.exists(torch_rccl):
        prelude = "import ctypes\nfrom pisces_amd import _native\nctypes.CDLL(%r)" % torch_rccl
        r = _comm_library_in_a_fresh_process(prelude)
        assert r == "OK mapped: " + torch_rccl, r
        r = _comm_library_in_a_fresh_process(prelude, {"PISCES_HIP_RCCL_PATH": system})
        assert r == "OK PISCES_HIP_RCCL_PATH: " + system, r
    # 3. nothing mapped, nothing named: the loader's path
    r = _comm_library_in_a_fresh_process("")
    assert r.startswith("OK default: "), r
