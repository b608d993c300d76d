"""Import-surface shim (oracle only): the reference imports cv2 on the hot path but never calls it there."""
