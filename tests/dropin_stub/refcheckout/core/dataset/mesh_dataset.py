class MeshLoader(object):
    origin = 'stub reference checkout'
