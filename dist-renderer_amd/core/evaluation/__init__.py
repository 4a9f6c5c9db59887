from .create_mesh import create_sdf_grid, create_sdf_grid_speedup, get_samples, infer_samples

__all__ = ['create_sdf_grid', 'create_sdf_grid_speedup', 'get_samples', 'infer_samples']
