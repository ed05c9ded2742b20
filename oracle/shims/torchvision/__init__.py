"""torchvision stand-in (TEST INFRASTRUCTURE ONLY)."""
