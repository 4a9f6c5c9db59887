import os, sys
sys.path.append(os.path.dirname(os.path.abspath(__file__)))
from mesh_dataset import *
from loader_single import *
from loader_multi_pmodata import *
