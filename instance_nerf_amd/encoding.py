"""``get_encoder`` factory with upstream's signature."""


def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=2048, align_corners=False, **kwargs):
    if encoding == "None":
        return (lambda x, **kw: x), input_dim
    if encoding == "hashgrid":
        from .gridencoder import GridEncoder
        enc = GridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim,
                          base_resolution=base_resolution, log2_hashmap_size=log2_hashmap_size,
                          desired_resolution=desired_resolution, gridtype="hash", align_corners=align_corners)
    elif encoding == "sphere_harmonics":
        from .shencoder import SHEncoder
        enc = SHEncoder(input_dim=input_dim, degree=degree)
    else:
        raise NotImplementedError(
            f"encoding '{encoding}' is outside the hot path (SURVEY.md section 2 rows 13-14); "
            "available: None, hashgrid, sphere_harmonics")
    return enc, enc.output_dim
