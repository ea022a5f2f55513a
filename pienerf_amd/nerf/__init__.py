"""Mirror of the parts of the reference's ``nerf`` package that sit on the simulate-and-render path."""
