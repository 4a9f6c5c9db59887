"""Driver-shaped test script: performs the import lists of the reference's run_single_shape.py:1-14 and
run_multi_pmodata.py:1-15 (same statements, same order; the two third-party imports the test box lacks are left out) and reports
where every name came from."""
import numpy as np
import os, sys
import torch

sys.path.append(os.path.dirname(os.path.abspath(__file__)))
from core.dataset import LoaderSingle
from core.inv_optimizer import optimize_single_view
from core.evaluation import *
from core.utils.render_utils import *
from core.utils.decoder_utils import load_decoder
from core.visualize.vis_utils import *
from core.visualize import Visualizer
from core.sdfrenderer import SDFRenderer
import pickle
from core.dataset import LoaderMultiPMO
from core.visualize.visualizer import print_loss_pack_color, Visualizer
from core.sdfrenderer import SDFRenderer_warp
from core.inv_optimizer import optimize_multi_view
from core.utils import pytorch_ssim

import inspect
import json


def where(obj):
    return os.path.abspath(inspect.getsourcefile(obj))


ev = Evaluator(None)
print(json.dumps({
    'argv': sys.argv[1:],
    'LoaderSingle': where(LoaderSingle), 'LoaderMultiPMO': where(LoaderMultiPMO), 'Visualizer': where(Visualizer),
    'project_points': where(project_points), 'Evaluator': where(Evaluator), 'pytorch_ssim': where(pytorch_ssim),
    'Evaluator.create_mesh': where(ev.create_mesh), 'Evaluator.decode_sdf': where(ev.decode_sdf),
    'optimize_single_view': where(optimize_single_view), 'optimize_multi_view': where(optimize_multi_view),
    'SDFRenderer': where(SDFRenderer), 'SDFRenderer_warp': where(SDFRenderer_warp), 'load_decoder': where(load_decoder),
    'downsize_camera_intrinsic': where(downsize_camera_intrinsic), 'get_tensor_from_camera': where(get_tensor_from_camera),
    'create_mesh_speedup': where(create_mesh_speedup),
    'default_arith': sys.modules[SDFRenderer.__module__].default_arith(),      # (this build's launcher option, not in the reference)
}))
