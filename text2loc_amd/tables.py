"""Dataset constants of KITTI360Pose the hot path depends on (class vocabulary, colour centres): product data, shared by
the packer, the dataset reader and the synthetic-data generator."""
import numpy as np

# Table values follow datapreparation/kitti360pose/utils.py:48-69 (class names, alphabetical) and
# :210-231 (8 fitted colour centres / their names; 'gray' appears twice in the reference).
KNOWN_CLASS = [
    "box", "bridge", "building", "fence", "garage", "guard rail", "lamp", "pad", "parking", "pole",
    "road", "sidewalk", "smallpole", "stop", "terrain", "traffic light", "traffic sign",
    "trash bin", "tunnel", "vegetation", "vending machine", "wall",
]
COLOR_NAMES = ["dark-green", "gray", "gray-green", "bright-gray", "gray", "black", "green", "beige"]
COLORS = np.array(
    [
        [47.2579917, 49.75368454, 42.4153065],
        [136.32696657, 136.95241796, 126.02741229],
        [87.49822126, 91.69058836, 80.14558512],
        [213.91030679, 216.25033052, 207.24611073],
        [110.39218852, 112.91977458, 103.68638249],
        [27.47505158, 28.43996795, 25.16840296],
        [66.65951839, 70.22342483, 60.20395996],
        [171.00852191, 170.05737735, 155.00130334],
    ]
) / 255.0

