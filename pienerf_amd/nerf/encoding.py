"""get_encoder with the reference's signature (nerf/encoding.py:40-77), limited to the two encoders of the render path."""
from ..gridencoder import GridEncoder
from ..shencoder import SHEncoder


def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                desired_resolution=2048, align_corners=False, **kwargs):
    if encoding == "sphere_harmonics":
        encoder = SHEncoder(input_dim=input_dim, degree=degree)
    elif encoding in ("hashgrid", "tiledgrid"):
        encoder = GridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim, base_resolution=base_resolution,
                              log2_hashmap_size=log2_hashmap_size, desired_resolution=desired_resolution,
                              gridtype="hash" if encoding == "hashgrid" else "tiled", align_corners=align_corners)
    else:
        raise NotImplementedError(f"encoding {encoding!r} is not on the simulate-and-render path (hashgrid / tiledgrid / sphere_harmonics only)")
    return encoder, encoder.output_dim
