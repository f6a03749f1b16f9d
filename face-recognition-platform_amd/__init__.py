"""MI355X-native replacement for the detect -> align -> embed -> match loop behind
backend/app/services/face_service.py of achiever04/face-recognition-platform.

The directory name carries the reference's name and is not a valid Python identifier;
import it through `frp_amd_loader` at the repo root (registers it as package `frp_amd`).
"""
__all__ = ["netspec", "weights", "native", "face_service", "face_api", "gallery", "dist", "camera_loop", "watchlist"]
