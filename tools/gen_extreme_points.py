"""Write the `extreme_points` field into a COCO instances file (the reference's tools/gen_coco_lsvr.py):

    python tools/gen_extreme_points.py annotations/instances_train2017.json annotations/instances_lsvr_train2017.json
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsnet_amd.data.extreme_points import add_extreme_points  # noqa: E402

if __name__ == '__main__':
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    print('annotations:', add_extreme_points(sys.argv[1], sys.argv[2]))
