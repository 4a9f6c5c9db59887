import os, sys
sys.path.append(os.path.dirname(os.path.abspath(__file__)))
from visualizer import Visualizer, print_loss_pack, print_loss_pack_color
from vis_utils import *
