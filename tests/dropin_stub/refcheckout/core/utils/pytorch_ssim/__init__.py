def loss_ssim(a, b):
    return [0.0]
