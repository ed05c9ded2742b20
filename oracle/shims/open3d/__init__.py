"""open3d stand-in (TEST INFRASTRUCTURE ONLY): empty module so `import open3d` in dataset readers succeeds."""
