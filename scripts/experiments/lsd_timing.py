"""Stage times of the LSD line path (plvs_hip_lsd_extract) beside the EDLines path on the same images."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import lsd_golden_scenario as S
from plvs_amd.lines import LineExtractor, LSDOptions


class Lsd(LineExtractor):
    skUseLsdExtractor = True


for name in S.IMAGES:
    img = S.image_of(dict(image=name))
    for cls in (Lsd, LineExtractor):
        ex = cls(100, LSDOptions(numOctaves=3, min_length=0.025, **dict(S.DEFAULTS, **S.TRACKING)))
        ex(img)
        ts, st = [], None
        for _ in range(10):
            t0 = time.perf_counter(); kl, d = ex(img); ts.append((time.perf_counter() - t0) * 1e3)
        print(name, cls.__name__, "lines", len(kl), "ms min %.2f median %.2f" % (min(ts), sorted(ts)[5]), {k: round(v, 2) for k, v in ex.stage_ms().items()})
        ex.close()
