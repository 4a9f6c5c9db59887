from core.visualize.vis_utils import project_points      # a reference package importing another reference package through `core`


class LoaderMultiPMO(object):
    helper = staticmethod(project_points)
