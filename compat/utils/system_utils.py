import os


def mkdir_p(folder_path):
    os.makedirs(folder_path, exist_ok=True)
