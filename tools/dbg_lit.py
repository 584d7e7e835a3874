import os, sys, importlib.util
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_special2.py")).read().split('if __name__ != "__main__":')[0]
ns = {"__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_special2.py"), "__name__": "fuzz_special2_head"}
exec(compile(src, "fuzz_special2_head", "exec"), ns)
build, prod, orc, scenes = ns["build"], ns["prod"], ns["orc"], ns["scenes"]
for s in [int(x) for x in sys.argv[1:]]:
    log = []
    build(orc, s, log=log)
    for flags in ((1, 1, 1), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)):
        got = scenes.render(build(prod, s, *[bool(f) for f in flags])); ref = scenes.render(build(orc, s, *[bool(f) for f in flags]))
        d = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max(axis=2)
        print("seed", s, "lights/2d/extras", flags, "beyond 1:", int((d > 1).sum()), "differ:", int((d > 0).sum()), "max", int(d.max()))
    for e in log:
        print("   ", e)
