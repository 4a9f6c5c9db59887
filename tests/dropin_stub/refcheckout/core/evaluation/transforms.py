from create_mesh import create_mesh, create_mesh_speedup      # flat: must land on the MI355X build's create_mesh
from decoder_utils import decode_sdf                          # flat: must land on the MI355X build's decode_sdf
