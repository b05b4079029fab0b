"""Regenerates tests/golden/maps/*.npy from the reference's map images.

Run in the build container only (it reads /root/reference/maps/*.png, which
does not exist on the GPU box).  Each output is the costmap_2d char map a
map_server + static layer would hand the planner: row 0 is the bottom image
row (map_server flips y), a pixel darker than occupied_thresh 0.65
(maps/map.yaml:4), i.e. mean(RGB) < 89.25, becomes 254 (lethal), everything
else 0.
"""
import sys
from pathlib import Path

import numpy as np
from PIL import Image

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/maps")
OUT = Path(__file__).parent / "maps"
OUT.mkdir(exist_ok=True)
for name in ["map_parking", "map_maze", "map_large", "map_basic", "map_small", "map_slalom", "map_dead_end"]:
    img = np.asarray(Image.open(REF / f"{name}.png").convert("RGB"), dtype=np.float64).mean(axis=2)
    occ = (255.0 - img) / 255.0 > 0.65
    charmap = np.where(occ, 254, 0).astype(np.uint8)[::-1].copy()
    np.save(OUT / f"{name}.npy", charmap)
    print(name, charmap.shape, float(occ.mean()))
