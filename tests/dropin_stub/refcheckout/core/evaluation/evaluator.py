from eval_func import *
from transforms import *


class Evaluator(object):
    def __init__(self, decoder):
        self.decoder, self.create_mesh, self.decode_sdf = decoder, create_mesh, decode_sdf
