"""Empty stand-in for OpenCV -- GOLDEN GENERATION ONLY: the reference's motion_filter.py / trajectory_filler.py import cv2 at
module scope and never use it on the paths the golden generators run."""
