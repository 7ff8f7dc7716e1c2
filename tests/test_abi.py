"""CPU-side checks of the drop-in boundary: the library loads and exports every symbol that
include/docodo_hip.h declares; host-only entry points behave; compute entry points fail loudly
(DCD_E_HIP) instead of falling back when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "docodo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dcd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import docodo_amd
    from docodo_amd import _lib
    L = docodo_amd.lib()
    names = _declared()
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), "libdocodo_hip.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, "python binding and header disagree"
    assert L.dcd_version() >= 1


def test_no_oracle_or_cpu_fallback_in_product():
    """The product path must never import the oracle."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "docodo_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                s = open(os.path.join(dp, f), errors="replace").read()
                assert "import oracle" not in s and "from oracle" not in s and "liboracle" not in s, f


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import docodo_amd
    from docodo_amd import _lib
    with pytest.raises(docodo_amd.DocodoError) as e:
        docodo_amd.IndexSequence.from_values([1, 2, 3])
    assert e.value.code == _lib.DCD_E_HIP
    with pytest.raises(docodo_amd.DocodoError) as e:
        docodo_amd.Builder()
    assert e.value.code == _lib.DCD_E_HIP


def test_synth_generator_host():
    from docodo_amd import synth
    t = synth.corpus_host(64)
    assert t.size == 64 * 1024
    docs = bytes(t).decode("ascii")
    assert set(docs) <= set("abcdefghijklmnopqrstuvwxyz ")
    # every document: whole words + single spaces, space padded to exactly 1024 bytes
    for d in range(64):
        doc = docs[d * 1024:(d + 1) * 1024]
        body = doc.rstrip(" ")
        assert "  " not in body and all(3 <= len(w) <= 10 for w in body.split(" "))
    assert docs[:1024] != docs[1024:2048]
    assert synth.term(1) == "aaaaaaa" and len({synth.term(k) for k in range(1, 5001)}) == 5000
    r = synth.query_ranks(100, 3)
    assert r.min() >= 101 and r.max() <= 100000 and all(len(set(x)) == 3 for x in r.tolist())
    assert np.array_equal(synth.corpus_host(8, first_doc=8), t[8 * 1024:16 * 1024])


def test_shift_is_host_side_and_matches_oracle(oracle_mod):
    """dcd_seq_shift re-codes the first value only (IndexSequence.cs:191-202); runs without a GPU."""
    import docodo_amd
    o = oracle_mod
    rng = np.random.default_rng(7)
    vals = np.cumsum(rng.integers(0, 0x10000, 100, dtype=np.uint64))
    u = o.seq_encode(np.concatenate([[0], vals]).astype(np.uint64))
    for s in (0, 100, 0xFFFFF, 0xFFFFFFFF, (1 << 61) + 5):
        q = docodo_amd.IndexSequence(u.copy())
        q.Shift(s)
        assert np.array_equal(q.units, o.seq_shift(u, s))


def _build_abi_demo(tmp_path):
    import subprocess
    exe = str(tmp_path / "abi_demo")
    libdir = os.path.join(ROOT, "docodo_amd")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "abi_demo.c"),
                    "-L", libdir, "-ldocodo_hip", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    return exe


def test_header_is_plain_c_and_the_demo_links(tmp_path):
    """include/docodo_hip.h compiles as C11 and examples/abi_demo.c links against the shared library: the boundary
    needs neither C++ nor Python nor torch on the caller's side.  Without a GPU the program fails loudly."""
    import subprocess
    import torch
    exe = _build_abi_demo(tmp_path)
    if not torch.cuda.is_available():
        r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
        assert r.returncode == 1 and "failed: status" in r.stderr     # DCD_E_HIP from the first compute call, no fallback


def test_hot_key_packing_host_unit(tmp_path):
    """tests/cpp/test_hot_key.cpp: the hot / cold sort key (dcd_sort.h rs_hot_key, dcd_codec.h hot_key_id) round-trips and keeps
    its order at every width the build may choose, and rs_hot_key_fits refuses the widths where 32 bits would wrap (255 hot
    terms over a cold id range of 2^25 .. 2^29: the build then sorts by plain ids).  Host code compiled by hipcc; runs on the CPU."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "test_hot_key")
    subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-pass-failed",
                    "-I", os.path.join(ROOT, "docodo_amd", "csrc"), "-I", os.path.join(ROOT, "include"), "-x", "hip",
                    os.path.join(ROOT, "tests", "cpp", "test_hot_key.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "hot key ok" in r.stdout, r.stdout[-2000:]


def test_build_flags_are_validated():
    """dcd_build_opts.flags was `reserved`: a caller that never zeroed it must not silently get a shard-only build."""
    import docodo_amd
    from docodo_amd import _lib
    o = _lib.BuildOpts()
    docodo_amd.lib().dcd_build_default_opts(C.byref(o))
    assert o.flags == 0
    o.flags = 0x40
    h = C.c_void_p()
    assert docodo_amd.lib().dcd_build_create(C.byref(o), C.byref(h)) == _lib.DCD_E_BADARG
    assert not h.value


@pytest.mark.gpu
def test_abi_demo_runs_from_plain_c(tmp_path):
    import subprocess
    exe = _build_abi_demo(tmp_path)
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("ok")
    assert os.path.exists(tmp_path / ".index") and os.path.exists(tmp_path / ".index.list")


def _rec(key, units):
    k = key.encode("utf-8")
    assert len(k) < 128
    return bytes([len(k)]) + k + len(units).to_bytes(4, "little") + b"".join(int(u).to_bytes(2, "little") for u in units)


@pytest.mark.parametrize("name,img", [
    ("short header", b"\x01\x02\x03"),
    # BinaryReader.Read7BitEncodedInt throws FormatException on a 6th length byte (Index.Load fails, Index.cs:373-376)
    ("six length bytes", (5).to_bytes(8, "little") + b"\x81\x80\x80\x80\x80\x00" + b"a" + (0).to_bytes(4, "little")),
    ("length runs off the image", (5).to_bytes(8, "little") + b"\x7f" + b"abc"),
    ("negative length", (5).to_bytes(8, "little") + b"\xff\xff\xff\xff\x0f" + b"abc"),
    ("truncated postings", (5).to_bytes(8, "little") + b"\x03abc" + (4).to_bytes(4, "little") + b"\x01\x00"),
    ("keys not ascending", (5).to_bytes(8, "little") + _rec("b", [1]) + _rec("a", [2])),
])
def test_malformed_index_images_are_io_errors(name, img):
    """The format checks of dcd_index_from_images are host-side and run before anything touches a device: a corrupt
    .index is DCD_E_IO with or without a GPU (the reference's Index.Load prints "Can't load" and returns false)."""
    import docodo_amd
    from docodo_amd import _lib
    with pytest.raises(docodo_amd.DocodoError) as e:
        docodo_amd.Index.from_images(img, b"")
    assert e.value.code == _lib.DCD_E_IO, name


@pytest.mark.gpu
def test_malformed_list_images_end_the_load_like_the_reference():
    """IndexPageList.Load swallows every exception and keeps the pages read so far (Build.cs:119-148): an empty string
    (s[0] throws), more than 5 length bytes (FormatException) and a truncated record all END the load at that record."""
    import docodo_amd
    idx = (10).to_bytes(8, "little") + _rec("abc", [3]) + _rec("abd", [7])

    def s(x):
        b = x.encode("utf-8")
        return bytes([len(b)]) + b

    good = (100).to_bytes(8, "little") + s("src:doc") + (100).to_bytes(8, "little") + s(":1") + (200).to_bytes(8, "little") + s(":2")
    more = (300).to_bytes(8, "little") + s(":3")
    for bad in ((250).to_bytes(8, "little") + b"\x00",                               # empty string
                (250).to_bytes(8, "little") + b"\x81\x80\x80\x80\x80\x00" + b"x",    # six length bytes
                (250).to_bytes(8, "little") + b"\x05:9"):                            # string runs off the image
        ix = docodo_amd.Index.from_images(idx, good + bad + (more if bad[-1:] != b"9" else b""))
        assert ix.n_pages == 2 and ix.n_docs == 1
        ix.close()
    ix = docodo_amd.Index.from_images(idx, good + more)
    assert ix.n_pages == 3
    ix.close()


def test_merge_index_files_is_host_side(tmp_path):
    """dcd_merge_index_files: streaming k-way merge of ".index" runs with disjoint keys (the out-of-core tail, MergeFiles
    Index.cs:577-689 for inputs that never share a key) -- host-only, runs without a GPU.  Header = max MaxCoord (Index.cs:596)."""
    import struct
    import docodo_amd
    from docodo_amd import _lib

    def rec(k, units):
        b = k.encode("utf-8")
        return bytes([len(b)]) + b + struct.pack("<i", len(units)) + np.array(units, np.uint16).tobytes()

    runs = [struct.pack("<Q", 50) + rec("abc", [1, 2]) + rec("zed", [9]) + rec("\u00e9t\u00e9", [5]),
            struct.pack("<Q", 70) + rec("abd", [3] * 70000) + rec("b", []),
            struct.pack("<Q", 10)]
    paths = []
    for i, x in enumerate(runs):
        p = tmp_path / ("run%d" % i)
        p.write_bytes(x)
        paths.append(str(p).encode())
    out = str(tmp_path / "out.index")
    arr = (C.c_char_p * len(paths))(*paths)
    _lib.check(docodo_amd.lib().dcd_merge_index_files(arr, len(paths), out.encode()))
    want = struct.pack("<Q", 70) + rec("abc", [1, 2]) + rec("abd", [3] * 70000) + rec("b", []) + rec("zed", [9]) + rec("\u00e9t\u00e9", [5])
    assert open(out, "rb").read() == want                       # ordinal (UTF-16 code unit) key order
    twice = (C.c_char_p * 2)(paths[0], paths[0])
    assert docodo_amd.lib().dcd_merge_index_files(twice, 2, out.encode()) == _lib.DCD_E_BADARG     # a key in two inputs
    missing = (C.c_char_p * 1)(str(tmp_path / "nope").encode())
    assert docodo_amd.lib().dcd_merge_index_files(missing, 1, out.encode()) == _lib.DCD_E_IO
    # the merged file is a loadable index image as far as the host-side format checks go
    with pytest.raises(docodo_amd.DocodoError) as e:
        docodo_amd.Index.from_images(open(out, "rb").read()[:-1], b"")
    assert e.value.code == _lib.DCD_E_IO
    # a failed merge leaves NO output (it is written as "<out>.tmp" and renamed on success) and no exception crosses the ABI:
    # a truncated payload, and a damaged key length that announces 2^31 - 1 key bytes
    out2 = tmp_path / "out2.index"
    for bad in (runs[0][:-1], struct.pack("<Q", 1) + b"\xff\xff\xff\xff\x07" + b"abc"):
        pb = tmp_path / "bad"
        pb.write_bytes(bad)
        arr2 = (C.c_char_p * 2)(paths[1], str(pb).encode())
        assert docodo_amd.lib().dcd_merge_index_files(arr2, 2, str(out2).encode()) == _lib.DCD_E_IO
        assert not out2.exists() and not (tmp_path / "out2.index.tmp").exists()


def test_struct_layouts_agree_with_the_library():
    """Hosts that declare the ABI's structs by hand (ctypes here, P/Invoke in bindings/csharp) must have the library's sizes:
    dcd_struct_size reports sizeof as the library was built, dcd_version the ABI revision the header names."""
    import docodo_amd
    from docodo_amd import _lib
    L = docodo_amd.lib()
    hdr = open(os.path.join(ROOT, "include", "docodo_hip.h")).read()
    assert L.dcd_version() == int(re.search(r"#define\s+DCD_ABI_VERSION\s+(\d+)", hdr).group(1))
    for which, cls in enumerate([_lib.BuildOpts, _lib.BuildStats, _lib.QueryOpts, _lib.ResultsView, _lib.TermMeta]):
        assert L.dcd_struct_size(which) == C.sizeof(cls), cls.__name__
    assert L.dcd_struct_size(5) == 0 and L.dcd_struct_size(-1) == 0


def _c_struct_fields(hdr, name):
    body = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        out.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", first)[-1])
        out += [re.findall(r"[A-Za-z_][A-Za-z0-9_]*", r)[-1] for r in rest]
    return out


def _cs_struct_fields(src, name):
    body = re.search(r"public\s+struct\s+%s\s*\{(.*?)\}" % name, src, flags=re.S).group(1)
    body = re.sub(r"//[^\n]*", "", body)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl.startswith("public"):
            continue
        first, *rest = decl.split(",")
        out.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", first)[-1])
        out += [r.strip() for r in rest]
    return out


def test_csharp_and_ctypes_structs_name_the_headers_fields_in_order():
    """No C# compiler in this image: the P/Invoke structs are checked textually -- same field names in the same order as
    include/docodo_hip.h (LayoutKind.Sequential then gives the C layout), and the same for the ctypes mirror."""
    from docodo_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "docodo_hip.h")).read()
    cs = open(os.path.join(ROOT, "bindings", "csharp", "Native.cs")).read()
    pairs = [("dcd_build_opts", "BuildOpts", _lib.BuildOpts), ("dcd_build_stats", "BuildStats", _lib.BuildStats),
             ("dcd_query_opts", "QueryOpts", _lib.QueryOpts), ("dcd_results_view", "ResultsView", _lib.ResultsView),
             ("dcd_term_meta", "TermMeta", _lib.TermMeta)]
    for cname, csname, cls in pairs:
        want = _c_struct_fields(hdr, cname)
        assert [f[0] for f in cls._fields_] == want, cname
        if re.search(r"public\s+struct\s+%s\b" % csname, cs):
            assert _cs_struct_fields(cs, csname) == want, csname
