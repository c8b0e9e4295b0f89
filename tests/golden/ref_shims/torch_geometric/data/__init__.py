class Data:  # names only on the model path
    pass


class Batch(Data):
    pass
