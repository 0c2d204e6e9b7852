"""What the flattener DECIDES for every patch the `-m gpu` suite renders, without a GPU: the suite is run with Patch.render / render_channels /
render_raw replaced by a recorder that notes srack_render_info's text (units, fused shapes, approx[...] — host-side flattening only) for the
patch and flags at hand and then stops the test.  Two builds of the library can be compared this way before a GPU is at hand: a change of
csrc/approx.cpp or flatten.cpp that makes a test's patch lose the kernel the test asserts shows up as a changed line.
usage: decisions_dryrun.py out.json            (then again with the other libsrack_hip.so in place, and `--diff a.json b.json`)
Only the FIRST render of each test is seen (the recorder cannot return frames the test would accept)."""
import inspect, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if len(sys.argv) > 1 and sys.argv[1] == "--diff":
    a, b = (json.load(open(p)) for p in sys.argv[2:4])
    changed = [k for k in sorted(set(a) | set(b)) if a.get(k) != b.get(k)]
    print(f"{len(a)} / {len(b)} tests recorded, {len(changed)} differ")
    for k in changed:
        print(" ", k)
        print("    <", (a.get(k) or ["-"])[-1][-260:])
        print("    >", (b.get(k) or ["-"])[-1][-260:])
    sys.exit(0)

import pytest
import srack_pkg

S = srack_pkg.load()
records, current = {}, [None]


class Recorded(BaseException):
    pass


def recorder(name):
    orig = getattr(S.Patch, name)
    sig = inspect.signature(orig)

    def fake(self, *args, **kwargs):
        flags = sig.bind(self, *args, **kwargs).arguments.get("flags", 0)
        try:
            self.kernel_source(flags)  # (flattens with these flags; a patch the generator does not cover — a reverb — is flattened all the same)
        except S.SrackError:
            pass
        try:
            text = self.info()
        except S.SrackError as e:
            text = f"error {e.code}"
        records.setdefault(current[0], []).append(f"{name} flags {flags}: {text}")
        raise Recorded()
    return fake


for n in ("render", "render_channels", "render_raw"):
    setattr(S.Patch, n, recorder(n))
S.device_count = lambda: 1


class Plugin:
    def pytest_runtest_setup(self, item):
        current[0] = item.nodeid


out = sys.argv[1]
pytest.main([os.path.join(ROOT, "tests"), "-m", "gpu", "-q", "--no-header", "-p", "no:cacheprovider", "--tb=no", "-o", "addopts="] + sys.argv[2:], plugins=[Plugin()])
json.dump(records, open(out, "w"), indent=0)
print(f"{len(records)} tests recorded -> {out}")
