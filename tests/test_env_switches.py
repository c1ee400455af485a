"""The product's environment switches: every MOLLYHIP_* variable that the package reads is documented in include/mollyhip.h, there are at most 25 of
them, and each is either a user-facing knob described there or set by a test.  A switch that comes back without a measurement behind it fails here."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _vars(paths):
    out = set()
    for p in paths:
        out |= set(re.findall(r"MOLLYHIP_[A-Z0-9_]+", open(p, errors="ignore").read()))
    return out


def test_every_switch_the_product_reads_is_documented_and_few():
    product = _vars(glob.glob(os.path.join(ROOT, "molly.jl_amd", "*.py")) + glob.glob(os.path.join(ROOT, "molly.jl_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "molly.jl_amd", "csrc", "*.hip")))
    header = _vars([os.path.join(ROOT, "include", "mollyhip.h")])
    assert product <= header, f"read by the product, not documented in include/mollyhip.h: {sorted(product - header)}"
    assert len(product) <= 25, sorted(product)
    tests = _vars(glob.glob(os.path.join(ROOT, "tests", "*.py")))
    bench = _vars([os.path.join(ROOT, "bench.py")] + [f for f in glob.glob(os.path.join(ROOT, "tools", "*")) if os.path.isfile(f)])
    # knobs nobody's test sets must at least be the documented user-facing / tooling ones
    user_facing = {"MOLLYHIP_DEBUG", "MOLLYHIP_XFER_TIMEOUT_MS", "MOLLYHIP_REUSE_RUN_FORCES", "MOLLYHIP_GHOST_MARGIN_PM", "MOLLYHIP_FORCE_DOMAIN", "MOLLYHIP_LIB_AB",
                   "MOLLYHIP_DBG_TIMES", "MOLLYHIP_DBG_DUMP"}
    orphan = product - tests - bench - user_facing
    assert not orphan, f"neither set by a test or tool nor a documented user knob: {sorted(orphan)}"
