// stub: cg::reduce is not used by the reference's rasterizer sources
