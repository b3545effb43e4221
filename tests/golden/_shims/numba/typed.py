class List(list):
    pass
