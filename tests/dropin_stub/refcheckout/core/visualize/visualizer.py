from vis_utils import project_points


def print_loss_pack(loss_pack, name):
    print('stub', name)


def print_loss_pack_color(loss_pack, name):
    print('stub', name)


class Visualizer(object):
    def __init__(self, img_hw, dmin=0.0, dmax=10.0):
        self.img_hw, self.data, self.losses, self.packs, self.resets = img_hw, {}, [], 0, 0

    def reset_data(self):
        self.data, self.resets = {}, self.resets + 1

    def add_data(self, name, src, mask=None):
        self.data[name] = src

    def add_loss_from_pack(self, pack):
        self.packs += 1

    def add_loss(self, loss):
        self.losses.append(float(loss))
