"""8-bit quantisation and image output, identical to the reference's save_image
(ray_tracer.py:606-612): clip to [0,1], x255, truncate toward zero."""
import numpy as np


def quantize_u8(image_array) -> np.ndarray:
    a = np.clip(np.asarray(image_array, dtype=np.float64), 0, 1)
    return (a * 255).astype(np.uint8)


def save_image(image_array, output_path):
    from PIL import Image
    Image.fromarray(quantize_u8(image_array)).save(output_path)
    print(f"Image saved to {output_path}")
