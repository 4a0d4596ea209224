import json, sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
import bench, deepgemm_ascend_amd as dga
from deepgemm_ascend_amd.harness import sweep
sweep.SHAPE_GROUP = [s for s in sweep.SHAPE_GROUP if s[2] % 16]
print(json.dumps(bench.shape_list_leg(dga), indent=1))
