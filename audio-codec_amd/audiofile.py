"""audiofile.py mirror (coder/audiofile.py:51-92)."""


class CodingParams:
    """Attribute bag shared by the file objects (coder/audiofile.py:51-53)."""
    pass


class AudioFile:
    def __init__(self, filename):
        self.filename = filename

    def OpenForReading(self):
        self.fp = open(self.filename, "rb")
        return self.ReadFileHeader()

    def OpenForWriting(self, codingParams):
        self.fp = open(self.filename, "wb")
        self.WriteFileHeader(codingParams)

    def Close(self, codingParams):
        self.fp.close()

    def ReadFileHeader(self):
        return CodingParams()

    def ReadDataBlock(self, codingParams):
        pass

    def WriteFileHeader(self, codingParams):
        pass

    def WriteDataBlock(self, data, codingParams):
        pass
