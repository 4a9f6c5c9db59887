def project_points(points):
    return points
