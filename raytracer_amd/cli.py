"""Command line front end with the reference's arguments (ray_tracer.py:615-644):

    python -m raytracer_amd.cli <scene_file> <output_image> [--width N] [--height N]
                                [--sequential] [--precision fp64|fp32] [--seed S]
                                [--numpy-stream] [--device D]

The renderer is the HIP backend (there is no CPU path here).  `--sequential` selects, like the
reference's flag of that name (ray_tracer.py:620,633-637), the semantics of its per-pixel
`render` — still computed on the GPU.  `--numpy-stream` reproduces `np.random.seed(S)` + the
unmodified `render_vectorized` (fp64, single GPU).
"""
import argparse

from .image import save_image
from .render import render_hip
from .scene import parse_scene_file, separate_objects


def main(argv=None):
    parser = argparse.ArgumentParser(description="MI355X ray tracer (HIP backend)")
    parser.add_argument("scene_file", type=str, help="Path to the scene file")
    parser.add_argument("output_image", type=str, help="Name of the output image file")
    parser.add_argument("--width", type=int, default=500, help="Image width")
    parser.add_argument("--height", type=int, default=500, help="Image height")
    parser.add_argument("--precision", choices=["fp64", "fp32"], default="fp64")
    parser.add_argument("--seed", type=int, default=1234, help="soft-shadow jitter seed (pixel-keyed)")
    parser.add_argument("--sequential", action="store_true", help="semantics of the reference's sequential renderer")
    parser.add_argument("--numpy-stream", action="store_true", help="jitter = the NumPy stream np.random.seed(seed) feeds the reference")
    parser.add_argument("--device", type=int, default=0)
    args = parser.parse_args(argv)

    camera, scene_settings, objects = parse_scene_file(args.scene_file)
    materials, surfaces, lights = separate_objects(objects)
    print(f"Scene loaded: {len(materials)} materials, {len(surfaces)} surfaces, {len(lights)} lights")
    print(f"Rendering {args.width}x{args.height} image...")
    image_array = render_hip(camera, scene_settings, materials, surfaces, lights, args.width, args.height,
                             precision=args.precision, jitter=("numpy" if args.numpy_stream else "pixel", args.seed),
                             device=args.device, sequential=args.sequential)
    save_image(image_array, args.output_image)


if __name__ == "__main__":
    main()
