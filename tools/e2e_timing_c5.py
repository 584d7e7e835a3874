import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy as np, rusterix_amd
from rusterix_amd import scenes
from run_configs import config
prod = rusterix_amd.load()
cfg = config(prod, "C5")
out = np.zeros(cfg.width * cfg.height * 4, np.uint8)
for _ in range(6):
    scenes.render(cfg, out)
    sys.stderr.write("---\n")
