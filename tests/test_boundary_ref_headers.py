"""Row b of SURVEY.md section 8 -- the drop-in boundary is link-alike, not look-alike.

Build-container-only check (skipped wherever /root/reference is absent, e.g. on the GPU box; nothing
of the reference is copied or shipped): the adaptor translation units libtsd_amd/host/adaptors/*.cc
are compiled, UNCHANGED, against libtsd's OWN headers (-I/root/reference/core/include, fmt header-only
from torch's include tree), and their symbol tables are compared with those of libtsd's own objects
compiled the same way:

  * every factory libtsd's filtre-rt.cc / polyphase.cc / ra.cc export (filtrage.hpp:1324,1367-1368,
    1376-1385,1428-1429,1585-1590,1610-1652,1968-1998,2029-2039) is DEFINED by the adaptors under the
    identical mangled name -- those three objects can be replaced wholesale;
  * filtre_fft / filtre_rif_fft (fourier.hpp:370, filtrage.hpp:1402-1403), which fourier.cc defines
    among much else, are defined too (gpu_filtre_fft.cc stands in for fourier.cc:737-990);
  * gpu_fftplan.cc defines NO libtsd symbol (it links beside an unmodified fourier.cc) and references
    the plug point tsd::fourier::fftplan_defaut (fourier.hpp:35);
  * every symbol libtsd's own test callers (tests/test-filtres.cc, tests/test-ra.cc) take from those
    three objects is provided by the adaptors;
  * the adaptors pull nothing from libtsd beyond what libtsd's remaining objects define (their
    undefined tsd:: symbols are all defined by objects that stay: tableau, frat, fenetres, rif-fen ...
    -- checked against the declarations by the compile itself; listed for INTEGRATION.md).
"""
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/core"
HOST = os.path.join(ROOT, "libtsd_amd", "host")
ADAPTORS = ["gpu_filtre_rt", "gpu_polyphase", "gpu_ra", "gpu_fftplan", "gpu_filtre_fft"]
REF_UNITS = ["src/filtrage/filtre-rt", "src/reechan/polyphase", "src/reechan/ra", "src/fourier/fourier",
             "tests/test-filtres", "tests/test-ra"]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("g++") is None,
                                reason="needs the reference tree and g++ (build container only)")


def _fmt_include():
    import torch
    inc = os.path.join(os.path.dirname(torch.__file__), "include")
    assert os.path.exists(os.path.join(inc, "fmt", "format.h")), "fmt headers not found next to torch"
    return inc


def _flags():
    return ["g++", "-std=c++20", "-O0", "-w", "-DFMT_HEADER_ONLY=1", "-DLIBTSD_USE_PNG=0", "-DLIBTSD_USE_FREETYPE=0",
            "-DLIBTSD_USE_GTKMM=0", f"-I{REF}/include", f"-I{_fmt_include()}"]


@pytest.fixture(scope="module")
def objs(tmp_path_factory):
    out = tmp_path_factory.mktemp("boundary")
    jobs = []
    for a in ADAPTORS:   # our TUs: libtsd's headers FIRST, then only the extension header root and the C ABI
        jobs.append((a, _flags() + [f"-I{HOST}/include_ext", f"-I{ROOT}/include", "-c", f"{HOST}/adaptors/{a}.cc",
                                    "-o", str(out / f"{a}.o")]))
    for u in REF_UNITS:  # libtsd's own objects: compiled where they lie, outputs in the pytest tmp dir only
        jobs.append((u, _flags() + ["-c", f"{REF}/{u}.cc", "-o", str(out / (os.path.basename(u) + ".ref.o"))]))

    def run(job):
        name, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        return name, r.returncode, r.stderr[-3000:]

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        for name, rc, err in ex.map(run, jobs):
            assert rc == 0, f"{name} does not compile against libtsd's headers:\n{err}"
    return out


def _syms(path, defined):
    flag = "--defined-only" if defined else "--undefined-only"
    out = subprocess.run(["nm", flag, str(path)], capture_output=True, text=True, check=True).stdout
    res = {}
    for line in out.splitlines():
        parts = line.split()
        if len(parts) >= 2:
            res[parts[-1]] = parts[-2]
    return res


def _demangle(names):
    if not names:
        return []
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines()


FACTORY = re.compile(r"^_ZN3tsd8filtrage\d+(filtre_|ligne_a_retard|decimateur|rif_delais|forme_polyphase|iforme_polyphase)")


def test_adaptors_compile_against_reference_headers(objs):
    for a in ADAPTORS:
        assert (objs / f"{a}.o").stat().st_size > 0


def test_factories_have_the_reference_mangled_names(objs):
    ours = {}
    for a in ADAPTORS:
        ours.update({k: v for k, v in _syms(objs / f"{a}.o", True).items() if v in "TW"})
    total = 0
    for unit in ("filtre-rt", "polyphase", "ra"):
        ref = {k for k, v in _syms(objs / f"{unit}.ref.o", True).items() if v in "TW" and FACTORY.match(k)}
        assert len(ref) >= 4, f"{unit}: factory symbols not found in the reference object"
        missing = sorted(ref - set(ours))
        assert not missing, f"{unit}.cc exports that the adaptors do not define:\n" + "\n".join(_demangle(missing))
        total += len(ref)
    assert total >= 37            # 21 + 12 + 4 at this revision of libtsd
    # the two fourier.cc definitions gpu_filtre_fft.cc replaces
    fourier = _syms(objs / "fourier.ref.o", True)
    wanted = [k for k in fourier if re.match(r"^_ZN3tsd7fourier10filtre_fftE", k) or re.match(r"^_ZN3tsd8filtrage14filtre_rif_fft", k)]
    assert len(wanted) == 3, _demangle(wanted)     # filtre_fft + filtre_rif_fft<float> / <cfloat>
    assert all(k in ours for k in wanted), _demangle([k for k in wanted if k not in ours])


def test_fftplan_hook_links_beside_unmodified_fourier_cc(objs):
    hook_def = {k for k, v in _syms(objs / "gpu_fftplan.o", True).items() if v in "TWBDR"}
    fourier_def = {k for k, v in _syms(objs / "fourier.ref.o", True).items() if v in "TBDR"}
    clash = sorted(hook_def & fourier_def)
    assert not clash, "gpu_fftplan.o redefines symbols of fourier.cc:\n" + "\n".join(_demangle(clash))
    und = _syms(objs / "gpu_fftplan.o", False)
    plug = [k for k in und if "fftplan_defaut" in k]
    assert plug and all(k in _syms(objs / "fourier.ref.o", True) for k in plug), "fftplan_defaut is not the reference's object"
    # no strong libtsd-namespace definition at all in the hook TU
    strong = [k for k, v in _syms(objs / "gpu_fftplan.o", True).items() if v == "T" and k.startswith("_ZN3tsd")]
    assert not strong, _demangle(strong)


def test_reference_callers_resolve_against_the_adaptors(objs):
    """What libtsd's own tests take from filtre-rt.o / polyphase.o / ra.o is all provided by us."""
    ours = {}
    for a in ADAPTORS:
        ours.update({k: v for k, v in _syms(objs / f"{a}.o", True).items() if v in "TW"})
    replaced = {}
    for unit in ("filtre-rt", "polyphase", "ra"):
        replaced.update({k: v for k, v in _syms(objs / f"{unit}.ref.o", True).items() if v in "TWBDR"})
    needed = set()
    for caller in ("test-filtres", "test-ra"):
        for k in _syms(objs / f"{caller}.ref.o", False):
            # symbols the caller can only get from the replaced objects: not inline code it also carries itself
            if k in replaced and k not in _syms(objs / f"{caller}.ref.o", True):
                needed.add(k)
    assert len(needed) >= 10, "the callers reference suspiciously few factory symbols"
    missing = sorted(k for k in needed if k not in ours)
    assert not missing, "libtsd's tests need, and the adaptors lack:\n" + "\n".join(_demangle(missing))


def test_what_the_adaptors_need_from_libtsd(objs, capsys):
    """The adaptors' undefined tsd:: symbols = libtsd code that must stay in the link.  None of them may
    be a symbol of the three replaced objects other than the ones the adaptors define themselves."""
    ours_def, ours_und = set(), set()
    for a in ADAPTORS:
        ours_def |= set(_syms(objs / f"{a}.o", True))
        ours_und |= set(_syms(objs / f"{a}.o", False))
    ext = sorted(k for k in ours_und - ours_def if k.startswith("_ZN3tsd") or k.startswith("_ZNK3tsd"))
    circular = []
    for unit in ("filtre-rt", "polyphase", "ra"):
        d = {k for k, v in _syms(objs / f"{unit}.ref.o", True).items() if v == "T"}
        circular += [k for k in ext if k in d]
    assert not circular, "the adaptors depend on objects they replace:\n" + "\n".join(_demangle(circular))
    with capsys.disabled():
        sys.stderr.write("\n[boundary] libtsd symbols the adaptors link against (stay in libtsd):\n  " +
                         "\n  ".join(_demangle(ext)) + "\n")
    assert any("design_rif_fen" in k for k in ext) and any("itrp_sinc" in k for k in ext)   # filtre_reechan's helpers
