from mesh_dataset import MeshLoader            # flat import of a sibling, as the reference's loaders do


class LoaderSingle(object):
    mesh_loader = MeshLoader
